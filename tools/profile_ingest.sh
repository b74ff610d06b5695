#!/bin/bash
# Round 5, first GPU session: the ingest kernels AS THEY ARE AT HEAD under rocprofv3 (VERDICT r4 "next" item 1a).
#   tools/profile_ingest.sh <tag>            (run through gpurun; everything lands under gpurun_out/<tag>_*)
# 1. the GPU suite; 2. gd_inflate_kernel / gd_inflate_crc_wave_kernel on two chr20 files' members (108 k members, the
# shape of profiles/r10w) written four ways (libdeflate 1 / 6, zlib 1, libdeflate 6 with aux tags): --kernel-trace --stats,
# then FETCH_SIZE / WRITE_SIZE in passes of their own; 3. one genome read (file -> BED) with its phases, a kernel trace and
# tools/timeline.py's view of it; 4. the same genome with realistic records (aux tags, level 6).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
T=${1:-r12a}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
LOG=$O/${T}.log
: > $LOG
echo "== pytest -m gpu" >> $LOG
( cd $R && timeout 900 python -X faulthandler -m pytest tests -m gpu -x -q > $O/${T}_pytest.txt 2>&1; tail -3 $O/${T}_pytest.txt >> $LOG )

LEN=63025520,63025520
run_inflate() {   # name, env...
  local name=$1; shift
  echo "== inflate $name" >> $LOG
  ( cd /tmp && env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_inflate_$name -o x -- python $R/tools/inflate_bench.py $LEN > $O/${T}_inflate_$name.txt 2>&1 )
  grep -h "members\|kernel" $O/${T}_inflate_$name.txt | tail -2 >> $LOG
  f=$(find $O/${T}_inflate_$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && grep -h "Name\|gd_inflate\|gd_bam" $f >> $LOG
  find $O/${T}_inflate_$name -name "*kernel_trace.csv" -delete
}
run_inflate ld1
run_inflate ld6 SYNTH_BAM_LEVEL=6
run_inflate z1 SYNTH_BAM_ZLIB=1
run_inflate ld6aux SYNTH_BAM_LEVEL=6 SYNTH_BAM_AUX=1
pmc_inflate() {
  local name=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && env INFLATE_BENCH_NO_ZLIB=1 "$@" rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/${T}_pmc_${name}_$c -o x -- python $R/tools/inflate_bench.py $LEN > $O/${T}_pmc_${name}_$c.txt 2>&1 )
  done
  python3 - $O/${T}_pmc_${name} >> $LOG <<'PY'
import csv, glob, os, sys
stem = sys.argv[1]
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    per = {}
    for f in glob.glob(os.path.join(stem + "_" + c, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c and ("gd_inflate" in r["Kernel_Name"]):
                per.setdefault(r["Kernel_Name"].split("(")[0], []).append(float(r["Counter_Value"]))
    for k, v in per.items():
        print("  %s %s KiB per dispatch: %s" % (c, k, [round(x) for x in v]))
PY
  for c in FETCH_SIZE WRITE_SIZE; do find $O/${T}_pmc_${name}_$c -name "*.csv" -size +2M -delete; done
}
echo "== pmc ld1" >> $LOG; pmc_inflate ld1
echo "== pmc ld6aux" >> $LOG; pmc_inflate ld6aux SYNTH_BAM_LEVEL=6 SYNTH_BAM_AUX=1

GENOME=$(python3 -c "import sys; sys.path.insert(0,'$R'); from goleft_amd import synth; print(','.join(str(x) for x in synth.HG19_LENGTHS))")
echo "== genome file -> BED (libdeflate 1, short records)" >> $LOG
python $R/tools/scope3.py --length $GENOME --name chrS --no-host --pause 8 --rocprof $O/${T}_scope3_trace > $O/${T}_scope3_genome.json 2>$O/${T}_scope3_genome.err
python3 -c "
import json; d=json.load(open('$O/${T}_scope3_genome.json')); v=d['device_decoder']; print('  wall %.3f s  %.3e ref-b/s  bam %.1f MB' % (v['wall_s'], v['ref_bases_per_s'], d['bam_MB'])); print('  phases', {k: v['phases'][k] for k in sorted(v['phases']) if k.startswith('lib_') or k in ('setup_s','read_s','compute_s','rows_s','feed_s','decode_s','begin_s')})" >> $LOG 2>&1
tr=$(find $O/${T}_scope3_trace -name "*kernel_trace.csv" | head -1)
if [ -n "$tr" ]; then
  python $R/tools/timeline.py $tr --slice-ms 100 > $O/${T}_scope3_timeline.txt 2>&1
  python $R/tools/timeline.py $tr --slice-ms 100 --json > $O/${T}_scope3_timeline.json 2>/dev/null
  head -3 $O/${T}_scope3_timeline.txt >> $LOG
  st=$(find $O/${T}_scope3_trace -name "*kernel_stats.csv" | head -1); [ -n "$st" ] && head -20 $st >> $LOG
  # the trace itself: launches of the ingest kernels only, gzip'd (a genome read is a few thousand launches)
  python3 - $tr $O/${T}_scope3_kernel_trace_ingest.csv <<'PY'
import csv, sys
rd = csv.DictReader(open(sys.argv[1]))
keep = ("Kernel_Name", "Start_Timestamp", "End_Timestamp", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Queue_Id", "Stream_Id")
cols = [c for c in keep if c in rd.fieldnames]
w = csv.writer(open(sys.argv[2], "w")); w.writerow(cols)
for r in rd:
    w.writerow([r[c].split("(")[0] if c == "Kernel_Name" else r[c] for c in cols])
PY
  rm -f $tr
fi
echo "== genome file -> BED (libdeflate 6, aux tags)" >> $LOG
SYNTH_BAM_LEVEL=6 SYNTH_BAM_AUX=1 python $R/tools/scope3.py --length $GENOME --name chrS --no-host --pause 8 > $O/${T}_scope3_genome_aux6.json 2>$O/${T}_scope3_genome_aux6.err
python3 -c "
import json; d=json.load(open('$O/${T}_scope3_genome_aux6.json')); v=d['device_decoder']; print('  wall %.3f s  %.3e ref-b/s  bam %.1f MB' % (v['wall_s'], v['ref_bases_per_s'], d['bam_MB'])); print('  phases', {k: v['phases'][k] for k in sorted(v['phases']) if k.startswith('lib_') or k in ('setup_s','read_s','compute_s','rows_s','feed_s','decode_s','begin_s')})" >> $LOG 2>&1
cat $LOG
