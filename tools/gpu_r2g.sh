#!/bin/bash
# round 2 evidence run: whole GPU suite, bench lines (wgs / chr20 / ont / cohort), rocprofv3 stats + PMC of the wgs command
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r2g_tests.log; cat gpurun_out/r2g_tests.log
timeout 600 python bench.py --verify > gpurun_out/r2g_bench_wgs.json 2> gpurun_out/r2g_wgs.err
timeout 300 python bench.py --workload chr20 --no-cpu-baseline --no-host-stream > gpurun_out/r2g_bench_chr20.json 2> gpurun_out/r2g_chr20.err
timeout 600 python bench.py --workload ont --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r2g_bench_ont.json 2> gpurun_out/r2g_ont.err
timeout 600 python bench.py --workload cohort --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2g_bench_cohort.json 2> gpurun_out/r2g_cohort.err
python - <<'P'
import json
for w in ("wgs","chr20","ont","cohort"):
    try:
        d=json.load(open('gpurun_out/r2g_bench_%s.json'%w))
        print(w, 'ms/step %.3f'%d['ms_per_step'], d['kernels_ms'], 'frac %.3f'%d['roofline']['frac'], 'value %.3e'%d['value'], d.get('verified_bit_exact'), d['config'].get('tiles_on_the_generic_kernel'), d['config'].get('ingest_normalise_ms_rank0'))
    except Exception as e: print(w, 'ERR', e)
P
bash tools/prof.sh r2g > gpurun_out/r2g_prof.log 2>&1
python tools/traffic_from_pmc.py gpurun_out/prof_r2g gpurun_out/r2g_wgs_traffic.json gd_tile_fast_kernel 2>&1 | tail -1
grep -A3 "kernel stats" gpurun_out/prof_r2g/summary.txt | head -8
