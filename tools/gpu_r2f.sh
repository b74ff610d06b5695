#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_normalize.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -5
python - <<'P'
import numpy as np
from goleft_amd import synth, engine as E
import torch
dev=torch.device('cuda',0)
names, lengths = list(synth.HG19_NAMES), list(synth.HG19_LENGTHS)
with E.DepthEngine(0) as eng:
    eng.set_params(window_size=1000)
    eng.set_contigs(lengths)
    keep=[]
    for t,L in enumerate(lengths):
        s=synth.short_reads_torch(L, synth.n_reads_for(L), t+1, dev); keep.append(s)
        eng.adopt_device(t,*s)
    eng.set_profiling(True)
    for fast in (1,0,1,0):
        eng.set_option(E.OPT_FAST_KERNEL,fast)
        ms=[]
        for i in range(6):
            eng.compute(); ms.append(eng.kernel_ms(E.K_TILE))
        st=eng.stats()
        print('fast',fast,'slow tiles',st.n_slow_tiles,'lookback',st.lookback,'tile ms min %.3f mean %.3f'%(min(ms[1:]),np.mean(ms[1:])), 'prep %.3f'%eng.kernel_ms(E.K_PREP))
P
