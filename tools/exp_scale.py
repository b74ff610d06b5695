"""Tile-kernel time vs job size on one device (is a rank's 1/8 shard as efficient per base as the whole genome?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from goleft_amd import synth, engine as E
dev = torch.device('cuda', 0)
L = list(synth.HG19_LENGTHS)
with E.DepthEngine(0) as eng:
    eng.set_params(window_size=1000); eng.set_contigs(L); keep = []
    for t, l in enumerate(L):
        s = synth.short_reads_torch(l, synth.n_reads_for(l), t + 1, dev); keep.append(s); eng.adopt_device(t, *s)
    eng.set_profiling(True)
    for fast, pb in ((1, 1), (1, 0), (0, 1)):
      eng.set_option(E.OPT_FAST_KERNEL, fast); eng.set_outputs(perbase=bool(pb)); print("fast", fast, "perbase", pb)
      for sel in ([20], [0], [0, 14, 20], list(range(8)), list(range(24))):
          eng.select_contigs(sel)
          ms = []
          for _ in range(8):
              eng.compute(); ms.append(eng.kernel_ms(E.K_TILE))
          nb = sum(L[t] for t in sel)
          print("contigs %-22s %6.1f Mb: tile min %.4f mean %.4f ms -> %.3f ms/Gb" % (str(sel)[:22], nb / 1e6, min(ms[2:]), np.mean(ms[2:]), min(ms[2:]) / nb * 1e9), flush=True)
