#!/bin/bash
# One GPU session on the inflate kernels (run through gpurun): the GPU inflate tests, then both kernels (GD_OPT_INFLATE_KERNEL 0: a
# lane per member, 1: a workgroup per member) on the members of two 30x chr20 files -- rocprofv3 --kernel-trace --stats, then
# FETCH_SIZE / WRITE_SIZE in passes of their own (--pmc is never combined with other trace domains).
#   tools/profile_inflate.sh <tag> [LEN] [env...]        everything lands under gpurun_out/<tag>_*
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; T=${1:-r13f}; LEN=${2:-63025520,63025520}; shift; shift
O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
LOG=$O/${T}.log; : > $LOG
echo "== pytest -m gpu tests/test_gpu_bamdecode.py" >> $LOG
( cd $R && timeout 1200 python -X faulthandler -m pytest tests/test_gpu_bamdecode.py -m gpu -x -q > $O/${T}_pytest.txt 2>&1; tail -3 $O/${T}_pytest.txt >> $LOG )
for k in 0 1; do
  echo "== kernel $k: trace" >> $LOG
  ( cd /tmp && env INFLATE_BENCH_KERNELS=$k "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_k${k} -o x -- python $R/tools/inflate_bench.py $LEN > $O/${T}_k${k}.txt 2>&1 )
  grep -h "members\|kernel " $O/${T}_k${k}.txt | tail -2 >> $LOG
  f=$(find $O/${T}_k${k} -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $O/${T}_k${k}_kernel_stats.csv && grep -h "Name\|gd_inflate" $f >> $LOG
  find $O/${T}_k${k} -name "*kernel_trace.csv" -delete
  for c in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && env INFLATE_BENCH_KERNELS=$k INFLATE_BENCH_NO_ZLIB=1 "$@" rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/${T}_pmc_k${k}_$c -o x -- python $R/tools/inflate_bench.py $LEN > $O/${T}_pmc_k${k}_$c.txt 2>&1 )
  done
  python3 - $O/${T}_pmc_k${k} >> $LOG <<'PY'
import csv, glob, os, sys
stem = sys.argv[1]
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    per = {}
    for f in glob.glob(os.path.join(stem + "_" + c, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c and ("gd_inflate" in r["Kernel_Name"]):
                per.setdefault(r["Kernel_Name"].split("(")[0], []).append(float(r["Counter_Value"]))
    for name, v in per.items():
        print("  %s %s KiB per dispatch: %s" % (c, name, [round(x) for x in v]))
PY
  for c in FETCH_SIZE WRITE_SIZE; do find $O/${T}_pmc_k${k}_$c -name "*.csv" -size +2M -delete; done
done
cat $LOG
