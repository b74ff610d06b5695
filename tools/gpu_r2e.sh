#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r2e -- python $R/bench.py --no-cpu-baseline --no-host-stream --steps 5 --warmup 2 > $R/gpurun_out/r2e_bench.json 2> $R/gpurun_out/r2e.err
cd $R
f=$(find gpurun_out/prof_r2e -name "*kernel_stats.csv" | head -1)
grep -E "gd::|Name" $f | cut -c1-200 | head -20
python - <<'P'
import json
d=json.load(open('gpurun_out/r2e_bench.json'))
print(d['ms_per_step'], d['kernels_ms'], d['roofline']['frac'])
P
python - <<'P'
import numpy as np
from goleft_amd import synth, engine as E
import torch
L=synth.CHR20_LEN
dev=torch.device('cuda',0)
s=synth.short_reads_torch(L, synth.n_reads_for(L), 20, dev)
with E.DepthEngine(0) as eng:
    eng.set_params(window_size=1000)
    eng.set_contigs([L]); eng.adopt_device(0,*s); eng.set_profiling(True)
    for i in range(3):
        eng.compute(); st=eng.stats(); print('slow tiles', st.n_slow_tiles, 'tiles', st.n_tiles, 'lookback', st.lookback, 'tile ms', eng.kernel_ms(E.K_TILE))
    eng.set_option(E.OPT_FAST_KERNEL,0)
    for i in range(2):
        eng.compute(); st=eng.stats(); print('generic: slow', st.n_slow_tiles, 'tile ms', eng.kernel_ms(E.K_TILE))
P
find gpurun_out/prof_r2e -name "*.csv" -size +2M -delete
