"""Step wall time (gd_compute, synchronous) with and without kernel-event profiling, chr20 and an N=8 shard."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from goleft_amd import synth, engine as E
dev = torch.device('cuda', 0)
for name, lengths in (("chr20", [synth.CHR20_LEN]), ("shard of 8 (chr1,15,21)", [synth.HG19_LENGTHS[0], synth.HG19_LENGTHS[14], synth.HG19_LENGTHS[20]])):
    with E.DepthEngine(0) as eng:
        eng.set_params(window_size=1000); eng.set_contigs(lengths); keep = []
        for t, L in enumerate(lengths):
            s = synth.short_reads_torch(L, synth.n_reads_for(L), t + 1, dev); keep.append(s); eng.adopt_device(t, *s)
        for prof in (0, 1, 0, 1):
            eng.set_profiling(bool(prof))
            for _ in range(3): eng.compute()
            t0 = time.perf_counter()
            for _ in range(50): eng.compute()
            dt = (time.perf_counter() - t0) / 50
            print("%s profiling %d: %.4f ms/step   kernels prep %.4f tile %.4f runs %.4f" % (name, prof, dt * 1e3, eng.kernel_ms(E.K_PREP), eng.kernel_ms(E.K_TILE), eng.kernel_ms(E.K_RUNS)), flush=True)
