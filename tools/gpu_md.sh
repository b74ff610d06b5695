#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
T=${1:-md}
{
echo "== pytest gpu multidepth + seqstats"; timeout 900 python -m pytest tests/test_multidepth.py tests/test_seqstats.py -m gpu -x -q 2>&1 | tail -15
echo "== scope iii: goleft depth --chrom chr1 -p 20 -o -w 16384 on a synthetic 30x chr1 BAM (indexcov/paper/cmp.sh:6)"
timeout 900 python tools/scope3.py --paper --name chr1 --length 249250621 2>gpurun_out/${T}_scope3.err | tail -1 | tee gpurun_out/${T}_scope3_chr1.json | cut -c1-1800
tail -3 gpurun_out/${T}_scope3.err
} > gpurun_out/md_$T.log 2>&1
cat gpurun_out/md_$T.log
bash tools/prof.sh r2i_ont --workload ont > /dev/null 2>&1
cat gpurun_out/prof_r2i_ont/summary.txt | grep -A20 "gd_ltile2" | head -24
