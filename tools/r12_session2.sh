#!/bin/bash
# Round 5, second GPU session: the coalesced CRC kernel, what the inflate kernel's time is made of (GD_OPT_INFLATE_PROBE),
# the CU split, and the whole new bench line.   tools/r12_session2.sh <tag>
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; T=${1:-r12b}; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
LOG=$O/${T}.log; : > $LOG
echo "== pytest inflate / BAM decode" >> $LOG
timeout 600 python -m pytest tests/test_gpu_bamdecode.py tests/test_gpu_ref_fixtures.py -m gpu -x -q > $O/${T}_pytest.txt 2>&1; grep -h "passed\|failed" $O/${T}_pytest.txt | tail -2 >> $LOG
LEN=63025520,63025520
echo "== inflate ld1, probes 0..3 (kernel-trace --stats)" >> $LOG
( cd /tmp && INFLATE_BENCH_PROBES=0,1,2,3 rocprofv3 --kernel-trace --output-format csv -d $O/${T}_inflate_probes -o x -- python $R/tools/inflate_bench.py $LEN > $O/${T}_inflate_probes.txt 2>&1 )
grep -h "lds pad\|kernel" $O/${T}_inflate_probes.txt >> $LOG
tr=$(find $O/${T}_inflate_probes -name "*kernel_trace.csv" | head -1)
python3 - $tr >> $LOG <<'PY'
import csv, sys
rows = [(r["Kernel_Name"].split("(")[0], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6) for r in csv.DictReader(open(sys.argv[1])) if "gd_inflate" in r["Kernel_Name"]]
print("  launches in order (ms):", [(n.replace("gd::gd_inflate_", ""), round(ms, 2)) for n, ms in rows])
PY
rm -f $tr
echo "== inflate ld6aux, probes 0,3" >> $LOG
( cd /tmp && SYNTH_BAM_LEVEL=6 SYNTH_BAM_AUX=1 INFLATE_BENCH_PROBES=0,3 INFLATE_BENCH_NO_ZLIB=1 python $R/tools/inflate_bench.py $LEN > $O/${T}_inflate_probes_ld6aux.txt 2>&1 )
grep -h "lds pad\|kernel" $O/${T}_inflate_probes_ld6aux.txt >> $LOG
for c in FETCH_SIZE; do
  ( cd /tmp && INFLATE_BENCH_NO_ZLIB=1 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/${T}_pmc_$c -o x -- python $R/tools/inflate_bench.py $LEN > $O/${T}_pmc_$c.txt 2>&1 )
  python3 - $O/${T}_pmc_$c $c >> $LOG <<'PY'
import csv, glob, os, sys
per = {}
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == sys.argv[2] and "gd_inflate" in r["Kernel_Name"]:
            per.setdefault(r["Kernel_Name"].split("(")[0], []).append(float(r["Counter_Value"]))
for k, v in per.items():
    print("  %s %s KiB per dispatch: %s" % (sys.argv[2], k, [round(x) for x in v]))
PY
  find $O/${T}_pmc_$c -name "*.csv" -size +2M -delete
done
GENOME=$(python3 -c "import sys; sys.path.insert(0,'$R'); from goleft_amd import synth; print(','.join(str(x) for x in synth.HG19_LENGTHS))")
echo "== genome file -> BED, CU split 8 (default) / 16 / 12" >> $LOG
python $R/tools/scope3.py --length $GENOME --name chrS --no-host --pause 8 --variants "GOLEFT_INGEST_CU_SPLIT=16;GOLEFT_INGEST_CU_SPLIT=12" > $O/${T}_scope3_genome.json 2>$O/${T}_scope3_genome.err
python3 -c "
import json; d=json.load(open('$O/${T}_scope3_genome.json'))
for k, v in d.items():
    if isinstance(v, dict) and 'wall_s' in v:
        print('  %s %s wall %.3f s  %.3e ref-b/s' % (k, v['env'], v['wall_s'], v['ref_bases_per_s'])); print('     ', {a: v['phases'][a] for a in sorted(v['phases']) if a.startswith('lib_') or a in ('setup_s','read_s','rows_s','decode_s','begin_s')})" >> $LOG 2>&1
echo "== python bench.py (defaults)" >> $LOG
( cd $R && timeout 1200 python bench.py 2>$O/${T}_bench.err | tail -1 > $O/${T}_bench_wgs_n1.json )
python3 -c "
import json; d=json.load(open('$O/${T}_bench_wgs_n1.json'))
print('  step %.3f ms value %.3e frac %.3f first %.3f' % (d['ms_per_step'], d['value'], d['roofline']['frac'], d['first_compute']['ratio_to_warm']))
b=d['bam_file_scope']; print('  bam_file_scope', {k: b.get(k) for k in ('file','wall_s','value','outputs_identical','oracle_identical','paper_invocation_s','error','synth_bam_s')})
print('  paper', b.get('paper_invocation'))
for n, v in (b.get('variants') or {}).items(): print('  variant', n, {k: v.get(k) for k in ('deflate','records','bam_bytes','device_wall_s','host_wall_s','outputs_identical','oracle_identical','error')})
for n, v in (d.get('other_workloads') or {}).items(): print('  other', n, {k: v.get(k) for k in ('ms_per_step','value','seconds_in_bench','error')}, (v.get('roofline') or {}).get('frac'), (v.get('first_compute') or {}).get('ratio_to_warm'), v.get('kernels_ms'))
print('  emu', {n: round(v['projected_speedup'],2) for n, v in d['emulated_sharding']['by_n_gpus'].items()})
" >> $LOG 2>&1
tail -3 $O/${T}_bench.err >> $LOG
cat $LOG
