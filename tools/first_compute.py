#!/usr/bin/env python3
"""Where a context's FIRST gd_compute spends its time: fresh context, the 30x genome adopted,
one compute; then a second and a third, and the same in a second context of the process (gd_compute_timing: prepare =
allocations + contig table, enqueue, wait).  It was this script, with per-call timers inside the enqueue, that found the
first device-to-host copy command of a process blocking the enqueue behind the running kernels."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from goleft_amd import synth
from goleft_amd.engine import DepthEngine, K_PREP, K_TILE, K_RUNS

names, lengths = list(synth.HG19_NAMES), list(synth.HG19_LENGTHS)
if len(sys.argv) > 1:
    lengths = lengths[:int(sys.argv[1])]
dev = torch.device("cuda", 0)
streams = []
for t, L in enumerate(lengths):
    streams.append(synth.short_reads_torch(L, synth.n_reads_for(L, 30.0), t + 1, dev))
torch.cuda.synchronize()
for rep in range(2):
    eng = DepthEngine(0)
    eng.set_params(window_size=1000, min_mapq=1, min_cov=4)
    eng.set_contigs(lengths)
    t0 = time.perf_counter()
    for t, s in enumerate(streams):
        eng.adopt_device(t, *s)
    print("context %d: adopt %.2f ms" % (rep, (time.perf_counter() - t0) * 1e3))
    for k in range(3):
        eng.set_profiling(True)
        t0 = time.perf_counter()
        eng.compute()
        w = time.perf_counter() - t0
        tm = eng.compute_timing()
        st = eng.stats()
        print("  compute %d: wall %.3f ms  prepare %.3f enqueue %.3f wait %.3f | prep %.3f tile %.3f runs %.3f | lookback %d slow %d reruns %d"
              % (k, w * 1e3, tm["prepare_s"] * 1e3, tm["enqueue_s"] * 1e3, tm["wait_s"] * 1e3, eng.kernel_ms(K_PREP), eng.kernel_ms(K_TILE),
                 eng.kernel_ms(K_RUNS), st.lookback, st.n_slow_tiles, st.reruns), flush=True)
    eng.close()
