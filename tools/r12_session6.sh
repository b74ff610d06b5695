#!/bin/bash
# Round 5, sixth GPU session: the streaming sums kernel with odd reads' ops carried in the queue item (A/B against the kernel
# of rounds 2-4 on one box, FETCH_SIZE of both), parity of the new kernel, and the first computes after the warm-up at
# context creation.   tools/r12_session6.sh <tag>
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; T=${1:-r12g}; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
LOG=$O/${T}.log; : > $LOG
echo "== parity of the product build (sums-only, cohort, soak)" >> $LOG
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_soak.py tests/test_depthwed.py -m gpu -q -k "sums or cohort or config4 or soak or depthwed" > $O/${T}_pytest.txt 2>&1; grep -h "passed\|failed" $O/${T}_pytest.txt | tail -2 >> $LOG; grep -h "^FAILED\|^ERROR" $O/${T}_pytest.txt | head >> $LOG
summ='import sys,json
d=json.loads(sys.stdin.read())
print("   step %.3f ms  kernel %.3f ms  frac %.3f (%.3f)  first %.3f ms (x%.3f, alloc %.2f ms)" % (d["ms_per_step"], d["roofline"]["avg_kernel_ms"], d["roofline"]["frac"], d["roofline"]["frac_bytes_really_read"], d["first_compute"]["ms"], d["first_compute"]["ratio_to_warm"], d["first_compute"]["prepare_alloc_ms"]))'
for v in base inline6 inline0 base6 inline6 base; do
  echo "== cohort, variant $v" >> $LOG
  GOLEFT_DEPTH_SO=$R/goleft_amd/variants/libgoleft_depth_$v.so timeout 600 python bench.py --workload cohort --steps 8 --warmup 2 --no-cpu-baseline 2>$O/${T}_cohort_$v.err | grep '^{"metric' | tail -1 > $O/${T}_bench_cohort_$v.json
  python3 -c "$summ" < $O/${T}_bench_cohort_$v.json >> $LOG 2>&1
done
for v in base inline6; do
  for c in FETCH_SIZE; do
    ( cd /tmp && GOLEFT_DEPTH_SO=$R/goleft_amd/variants/libgoleft_depth_$v.so rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/${T}_pmc_${v}_$c -o x -- python $R/bench.py --workload cohort --steps 2 --warmup 1 --no-cpu-baseline > $O/${T}_pmc_${v}_$c.txt 2>&1 )
    python3 - $O/${T}_pmc_${v}_$c $c $v >> $LOG <<'PY'
import csv, glob, os, sys
vals = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == sys.argv[2] and "gd_sums_stream_kernel" in r["Kernel_Name"]:
            vals.append(float(r["Counter_Value"]))
print("  %s %s gd_sums_stream_kernel KiB per dispatch: %s  (x2 x1024 = %.1f GB)" % (sys.argv[3], sys.argv[2], [round(x) for x in vals], (sum(vals) / max(1, len(vals))) * 2048 / 1e9))
PY
    find $O/${T}_pmc_${v}_$c -name "*.csv" -size +2M -delete
  done
done
echo "== first computes with the warm-up at context creation: chr20, ont (product build)" >> $LOG
for w in chr20 ont; do
  timeout 600 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-host-stream --emulate-shards= 2>$O/${T}_$w.err | grep '^{"metric' | tail -1 > $O/${T}_bench_${w}_n1.json
  python3 -c "$summ" < $O/${T}_bench_${w}_n1.json >> $LOG 2>&1
done
cat $LOG
