#!/bin/bash
# Round 5, third GPU session: the whole GPU suite on the pruned build with the fused long-read index, the long-read bench line,
# and where the inflate kernel's wave cycles go (SQ counters).   tools/r12_session3.sh <tag>
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; T=${1:-r12c}; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
LOG=$O/${T}.log; : > $LOG
echo "== pytest long-read structures first" >> $LOG
timeout 600 python -m pytest tests/test_gpu_longread.py -m gpu -x -q > $O/${T}_pytest_longread.txt 2>&1; grep -h "passed\|failed\|Error\|assert" $O/${T}_pytest_longread.txt | tail -6 >> $LOG
echo "== pytest -m gpu (all)" >> $LOG
timeout 1200 python -X faulthandler -m pytest tests -m gpu -q --maxfail=10 > $O/${T}_pytest.txt 2>&1; grep -h "passed\|failed" $O/${T}_pytest.txt | tail -3 >> $LOG; grep -h "^FAILED\|^ERROR" $O/${T}_pytest.txt | head -20 >> $LOG
echo "== bench ont" >> $LOG
timeout 900 python bench.py --workload ont --steps 10 --warmup 3 2>$O/${T}_ont.err | tail -1 > $O/${T}_bench_ont_n1.json
python3 -c "
import json; d=json.load(open('$O/${T}_bench_ont_n1.json'))
print('  step %.3f ms value %.3e frac %.3f kernels %s first %s' % (d['ms_per_step'], d['value'], d['roofline']['frac'], d['kernels_ms'], {k: d['first_compute'][k] for k in ('ms','wall_ms','prepare_alloc_ms','enqueue_ms','ratio_to_warm','kernels_ms')}))" >> $LOG 2>&1
tail -2 $O/${T}_ont.err >> $LOG
echo "== ont kernel trace" >> $LOG
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_ont_trace -o x -- python $R/bench.py --workload ont --steps 5 --warmup 2 --no-cpu-baseline --emulate-shards= > $O/${T}_ont_trace.txt 2>&1 )
f=$(find $O/${T}_ont_trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -h "Name\|gd::" $f | head -12 >> $LOG
find $O/${T}_ont_trace -name "*kernel_trace.csv" -delete
echo "== inflate kernel, SQ counters (ld1, 108 k members)" >> $LOG
LEN=63025520,63025520
i=1
for pmc in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_WAIT_INST_VMEM SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU"; do
  ( cd /tmp && INFLATE_BENCH_NO_ZLIB=1 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $O/${T}_sq$i -o x -- python $R/tools/inflate_bench.py $LEN > $O/${T}_sq$i.txt 2>&1 )
  python3 - $O/${T}_sq$i >> $LOG <<'PY'
import csv, glob, os, sys
per = {}
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "gd_inflate_kernel" in r["Kernel_Name"]:
            per.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k, v in sorted(per.items()):
    print("  %-24s %.4g per dispatch (%d dispatches)" % (k, sum(v) / len(v), len(v)))
PY
  find $O/${T}_sq$i -name "*.csv" -size +2M -delete
  i=$((i+1))
done
cat $LOG
