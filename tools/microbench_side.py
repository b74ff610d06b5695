#!/usr/bin/env python3
"""Kernel-only timings (HIP events inside the library) of the two side kernels:
gd_seq_stats (--stats) on a 249 Mb random sequence, W = 250 / 1000, and gd_md_flags
(multidepth) over S per-base vectors of a synthetic 30x chr20."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from goleft_amd import synth
from goleft_amd.engine import DepthEngine, K_SEQSTATS, K_MDFLAGS

rng = np.random.default_rng(1)
L = 249_250_621
seq = rng.choice(np.frombuffer(b"ACGTacgtNn", np.uint8), size=L)
with DepthEngine(0) as eng:
    eng.set_profiling(True)
    eng.seq_load(seq)
    for W in (1000, 250):
        st = np.arange(0, L, W, dtype=np.int64)
        en = np.minimum(st + W, L)
        eng.seq_stats(st, en)
        gc, cpg, low = eng.seq_stats(st, en)
        ms = eng.kernel_ms(K_SEQSTATS)
        print("seq_stats L=%d W=%d: %.3f ms, %.0f GB/s of bases (1 B/base + 12 B/window), gc=%.4f" % (
            L, W, ms, (L + 12 * len(st)) / ms / 1e6, gc.sum() / L))
S, Lc = 16, synth.CHR20_LEN
with DepthEngine(0) as eng:
    eng.set_params(window_size=1000, min_mapq=10, min_cov=4)
    eng.set_contigs([Lc] * S)
    n = synth.n_reads_for(Lc)
    base = synth.short_reads_numpy(Lc, n, 20)
    for s in range(S):
        eng.push(s, *base)                                  # the same stream S times: timing only
    eng.compute()
    eng.set_profiling(True)
    eng.md_flags(list(range(S)), 7, S // 2)
    a, f = eng.md_flags(list(range(S)), 7, S // 2)
    ms = eng.kernel_ms(K_MDFLAGS)
    print("md_flags S=%d L=%d: %.3f ms, %.0f GB/s (4*S B/position read)" % (S, Lc, ms, 4.0 * S * Lc / ms / 1e6))
    print("printed %.4f sufficient %.4f" % (a.mean(), f.mean()))
