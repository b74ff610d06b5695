"""Experiments on the WGS tile kernel (one box): python tools/exp_k1.py "<fast>,<perbase>,<dbg>" ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sys
import numpy as np
import torch
from goleft_amd import synth, engine as E
dev = torch.device('cuda', 0)
lengths = list(synth.HG19_LENGTHS)
with E.DepthEngine(0) as eng:
    eng.set_params(window_size=1000)
    eng.set_contigs(lengths)
    keep = []
    for t, L in enumerate(lengths):
        s = synth.short_reads_torch(L, synth.n_reads_for(L), t + 1, dev); keep.append(s)
        eng.adopt_device(t, *s)
    eng.set_profiling(True)
    eng.compute(); eng.compute()
    for v in sys.argv[1:]:
        fast, pb, dbg = [int(x) for x in v.split(",")]
        eng.set_option(E.OPT_FAST_KERNEL, fast)
        eng.set_outputs(perbase=bool(pb))
        eng.set_option(99, dbg)
        ms = []
        for i in range(5):
            eng.compute(); ms.append(eng.kernel_ms(E.K_TILE))
        print('fast %d perbase %d dbg %d: tile ms min %.3f mean %.3f  (slow tiles %d)' % (fast, pb, dbg, min(ms[1:]), np.mean(ms[1:]), eng.stats().n_slow_tiles), flush=True)
