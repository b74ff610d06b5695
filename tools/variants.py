#!/usr/bin/env python3
"""A/B timing of tile-kernel variants on ONE resident synthetic WGS stream.

    python tools/variants.py [--steps 8] [--verify] "KEY=VAL,KEY=VAL" ...

Each positional argument is one variant: a comma separated list of GOLEFT_GD_*
environment settings read by gd_create (TILE, THREADS, KERNEL=v6, OPT=0, PATH ...;
the GOLEFT_GD_ prefix is implied).  "-" is the default configuration.
The record streams are generated once on the device and adopted zero-copy by
a fresh engine per variant.  --verify checks chr21 against the CPU oracle.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("variants", nargs="*", default=["-"])
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--verify", action="store_true")
    ap.add_argument("--contigs", type=int, default=24)
    ap.add_argument("--window", type=int, default=1000)
    args = ap.parse_args()

    import torch
    from goleft_amd import synth
    from goleft_amd.engine import DepthEngine, K_PREP, K_TILE, K_RUNS

    dev = torch.device("cuda", 0)
    lengths = synth.HG19_LENGTHS[:args.contigs]
    streams = {}
    n_reads = n_ops = 0
    for t, L in enumerate(lengths):
        n = synth.n_reads_for(L)
        streams[t] = synth.short_reads_torch(L, n, t + 1, dev)
        n_reads += n
        n_ops += int(streams[t][4].shape[0])
    torch.cuda.synchronize()
    W = args.window
    n_win = sum((L + W - 1) // W for L in lengths)
    alg = synth.algorithmic_bytes(n_reads, n_ops, sum(lengths), n_win)

    want = None
    vt = min(20, len(lengths) - 1)       # chr21 when all contigs are present
    if args.verify:
        from oracle import pyoracle as po
        a = [x.cpu().numpy() for x in streams[vt]]
        r = po.Reads(a[0], a[1].view(np.uint16), a[2], a[3].view(np.uint32), a[4].view(np.uint32))
        want = po.perbase_c(r, 1, 0, lengths[vt], diff=True)

    for v in args.variants:
        for k in [k for k in os.environ if k.startswith("GOLEFT_GD_")]:
            del os.environ[k]
        if v != "-":
            for kv in v.split(","):
                k, val = kv.split("=")
                os.environ["GOLEFT_GD_" + k] = val
        eng = DepthEngine(0)
        eng.set_params(window_size=W, min_mapq=1, min_cov=4)
        eng.set_contigs(lengths)
        for t in streams:
            eng.adopt_device(t, *streams[t])
        eng.set_profiling(True)
        for _ in range(args.warmup):
            eng.compute()
        ms = {K_PREP: [], K_TILE: [], K_RUNS: []}
        for _ in range(args.steps):
            eng.compute()
            for k in ms:
                ms[k].append(eng.kernel_ms(k))
        st = eng.stats()
        tile = float(np.mean(ms[K_TILE]))
        line = "variant %-28s tile %.3f ms (min %.3f)  prep %.3f  runs %.3f  lookback %d  frac %.3f" % (
            v, tile, float(np.min(ms[K_TILE])), float(np.mean(ms[K_PREP])), float(np.mean(ms[K_RUNS])),
            st.lookback, alg / (tile * 1e-3) / 8e12)
        if want is not None:
            got = eng.perbase(vt)
            s, m = eng.windows(vt)
            ok = bool(np.array_equal(got, want))
            ok_w = bool(np.array_equal(s, np.add.reduceat(want.astype(np.int64), np.arange(0, len(want), W))))
            ok_m = bool(np.array_equal(m, np.minimum.reduceat(want, np.arange(0, len(want), W))))
            line += "  exact perbase=%s sums=%s mins=%s" % (ok, ok_w, ok_m)
        print(line, flush=True)
        eng.close()


if __name__ == "__main__":
    main()
