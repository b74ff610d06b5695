#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r2i -- python $R/tools/exp_step.py > $R/gpurun_out/r2i.log 2>&1
cd $R
f=$(find gpurun_out/prof_r2i -name "*kernel_stats.csv" | head -1)
grep -E "gd::|Name" $f | cut -c1-160
t=$(find gpurun_out/prof_r2i -name "*kernel_trace.csv" | head -1)
python - "$t" <<'P'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'gd::' in r['Kernel_Name'] and 'norm' not in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last 12 kernels: timeline
base=int(rows[-12]['Start_Timestamp'])
for r in rows[-12:]:
    print('%-40s start %8.1f us  dur %8.1f us'%(r['Kernel_Name'][:40], (int(r['Start_Timestamp'])-base)/1e3, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3))
P
find gpurun_out/prof_r2i -name "*.csv" -size +1M -delete
