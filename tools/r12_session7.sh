#!/bin/bash
# Round 5, seventh GPU session: the deletion-list pass with the next read's first ops asked for in advance, the contig table
# uploaded by a kernel -- parity first, then the long-read and chr20 lines.   tools/r12_session7.sh <tag>
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; T=${1:-r12h}; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
LOG=$O/${T}.log; : > $LOG
echo "== long-read parity" >> $LOG
timeout 900 python -m pytest tests/test_gpu_longread.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -q -k "longread or config5 or ont or long or chunk" > $O/${T}_pytest_long.txt 2>&1; grep -h "passed\|failed" $O/${T}_pytest_long.txt | tail -2 >> $LOG; grep -h "^FAILED\|^ERROR" $O/${T}_pytest_long.txt | head >> $LOG
summ='import sys,json
d=json.loads(sys.stdin.read())
print("   step %.3f ms  kernel %.3f ms  frac %.3f (%.3f)  first %.3f ms (x%.3f, alloc %.2f ms) kernels %s first-kernels %s" % (d["ms_per_step"], d["roofline"]["avg_kernel_ms"], d["roofline"]["frac"], d["roofline"]["frac_bytes_really_read"], d["first_compute"]["ms"], d["first_compute"]["ratio_to_warm"], d["first_compute"]["prepare_alloc_ms"], d["kernels_ms"], d["first_compute"]["kernels_ms"]))'
for w in ont chr20 ont chr20; do
  echo "== bench $w" >> $LOG
  timeout 600 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-host-stream --emulate-shards= 2>$O/${T}_$w.err | grep '^{"metric' | tail -1 > $O/${T}_bench_${w}_n1.json
  python3 -c "$summ" < $O/${T}_bench_${w}_n1.json >> $LOG 2>&1
done
echo "== ont kernel trace" >> $LOG
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_ont_trace -o x -- python $R/bench.py --workload ont --steps 5 --warmup 2 --no-cpu-baseline --emulate-shards= > $O/${T}_ont_trace.txt 2>&1 )
f=$(find $O/${T}_ont_trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -h "Name\|gd::" $f | head -8 >> $LOG
find $O/${T}_ont_trace -name "*kernel_trace.csv" -delete
echo "== pytest -m gpu (all)" >> $LOG
timeout 1200 python -X faulthandler -m pytest tests -m gpu -q --maxfail=10 > $O/${T}_pytest.txt 2>&1; grep -h "passed\|failed" $O/${T}_pytest.txt | tail -3 >> $LOG; grep -h "^FAILED\|^ERROR" $O/${T}_pytest.txt | head -20 >> $LOG
cat $LOG
