"""-m gpu: randomized soak of the engine against the CPU oracle.

One context lives through a whole sequence of jobs whose contig table, record mix,
parameters, device algorithm, output mode and tuning switches change from job to job, so
what is checked is not only each kernel but everything the context carries between jobs
(canonical CIGARs, position index, deletion lists and tile index, per-tile tables, the
rerun state).  GOLEFT_SOAK_JOBS=<n> / GOLEFT_SOAK_SEEDS=<n> lengthen the run (default 4 seeds x 10 jobs).
"""
import os

import numpy as np
import pytest

from tests import helpers as H
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu

JOBS = int(os.environ.get("GOLEFT_SOAK_JOBS", "10"))
# GOLEFT_SOAK_ONLY="13,14": replay a seed's job sequence but hand only these jobs to the engine
ONLY = [int(x) for x in os.environ.get("GOLEFT_SOAK_ONLY", "").split(",") if x]


def _split_push(eng, rng, tid, r, live=True):
    """Push a contig's records in 1..3 consecutive batches."""
    n = len(r.pos)
    cuts = sorted(set([0, n] + [int(x) for x in rng.integers(0, n + 1, size=int(rng.integers(0, 3)))]))
    for a, b in zip(cuts[:-1], cuts[1:]):
        if a == b or not live:
            continue
        o0, o1 = int(r.cigar_off[a]), int(r.cigar_off[b])
        eng.push(tid, r.pos[a:b], r.flag[a:b], r.mapq[a:b], (r.cigar_off[a:b + 1] - o0).astype(np.uint32),
                 r.cigar[o0:o1])


def _gappy_reads(rng, L, n):
    """Spliced-looking records: M blocks separated by N skips and D runs of up to several
    tiles, some running off the contig end."""
    pos = np.sort(rng.integers(0, max(1, L), size=n)).astype(np.int32)
    offs, ops = [0], []
    for _ in range(n):
        k = int(rng.integers(1, 7))
        for j in range(k):
            ops.append((int(rng.integers(1, 200)) << 4) | 0)
            if j + 1 < k:
                ops.append((int(rng.choice([1, 30, 4095, 4096, 4097, 9000, 70000])) << 4) | int(rng.choice([2, 3])))
        if rng.random() < 0.2:
            ops.append((int(rng.integers(1, 50)) << 4) | int(rng.choice([4, 5, 2, 3])))      # trailing clip / skip
        offs.append(len(ops))
    flag = rng.choice([0, 16, 0x400, 0x100], size=n, p=[0.5, 0.4, 0.05, 0.05]).astype(np.uint16)
    mapq = rng.choice([0, 3, 60], size=n, p=[0.1, 0.1, 0.8]).astype(np.uint8)
    return po.Reads(pos, flag, mapq, np.asarray(offs, np.uint32), np.asarray(ops, np.uint32))


def _contig_reads(rng, L):
    from goleft_amd import synth
    kind = rng.choice(["none", "random", "randlong", "short", "ont", "longcigar", "gappy", "pile"],
                      p=[0.08, 0.2, 0.12, 0.2, 0.1, 0.1, 0.12, 0.08])
    if kind == "none" or L == 0:
        return None
    if kind == "random":
        return H.random_reads(rng, L, int(rng.integers(0, 5000)))
    if kind == "randlong":
        return H.random_reads(rng, L, int(rng.integers(0, 3000)), max_ops=int(rng.integers(1, 40)), long_reads=True)
    if kind == "short":
        n = max(1, int(L * float(rng.choice([0.5, 5, 30, 80])) / 150))
        return po.Reads(*synth.short_reads_numpy(L, n, int(rng.integers(1, 1 << 30))))
    if kind == "ont":
        n = max(1, synth.n_ont_reads_for(L))
        return po.Reads(*synth.ont_reads_numpy(L, n, int(rng.integers(1, 1 << 30))))
    if kind == "longcigar":
        n_ops = [int(x) for x in rng.choice([1, 2, 23, 24, 25, 63, 64, 65, 129, 700, 4097], size=int(rng.integers(1, 60)))]
        return H.long_cigar_reads(rng, L, n_ops, max_step=int(rng.choice([3, 12, 40])),
                                  skip_every=int(rng.choice([0, 3])))
    if kind == "gappy":
        return _gappy_reads(rng, L, int(rng.integers(1, 3000)))
    n = int(rng.integers(1000, 60000))                                  # a pile on a few bases
    a = int(rng.integers(0, L))
    pos = np.sort(rng.integers(a, min(L, a + 40) + 1, size=n)).astype(np.int32)
    off = np.arange(n + 1, dtype=np.uint32)
    return po.Reads(pos, np.zeros(n, np.uint16), np.full(n, 60, np.uint8), off,
                    np.full(n, (int(rng.integers(1, 400)) << 4), np.uint32))


@pytest.mark.parametrize("seed", range(int(os.environ.get("GOLEFT_SOAK_SEEDS", "4"))))
def test_soak_one_context_many_jobs(seed):
    from goleft_amd.engine import (DepthEngine, GdError, PATH_AUTO, PATH_TILE, PATH_SCATTER, PATH_CHUNK,
                                   OPT_NT_STORES, OPT_NORMALIZE, OPT_FAST_KERNEL)
    rng = np.random.default_rng(9000 + seed)
    with DepthEngine(0) as eng:
        for job in range(JOBS):
            n_ctg = int(rng.integers(1, 6))
            lens = [int(rng.choice([0, 1, 63, 64, 4095, 4096, 4097, 8192, 12289, int(rng.integers(1, 300000)),
                                    int(rng.integers(1, 300000))])) for _ in range(n_ctg)]
            reads = {}
            for t, L in enumerate(lens):
                r = _contig_reads(rng, L)
                if r is not None:
                    reads[t] = r
            W = int(rng.choice([1, 2, 31, 32, 33, 64, 100, 250, 1000, 4096, 5000, 100000]))
            Q = int(rng.choice([0, 1, 4, 61]))
            mincov = int(rng.integers(0, 9))
            maxmean = int(rng.choice([0, 0, 15, 2000]))
            step = W * int(rng.integers(1, 200)) if rng.random() < 0.5 else 0
            path = int(rng.choice([PATH_AUTO, PATH_AUTO, PATH_TILE, PATH_CHUNK, PATH_SCATTER]))
            mode = str(rng.choice(["full", "full", "windows", "sums"]))
            if path == PATH_SCATTER:
                mode = "full"            # the scatter path needs the per-base vector (GD_E_INVAL otherwise)
            opts = (int(rng.integers(0, 2)), int(rng.random() < 0.8), int(rng.random() < 0.8))
            live = not ONLY or job in ONLY
            if not live:                                     # keep the generator in step, skip the engine
                for t, r in reads.items():
                    _split_push(eng, rng, t, r, live=False)
                rng.integers(1, 3)
                for t, L in enumerate(lens):
                    if mode == "full" and L > 2 and rng.random() < 0.5:
                        rng.integers(int(rng.integers(0, L)), L + 1)
                continue
            eng.set_option(OPT_NT_STORES, opts[0])
            eng.set_option(OPT_NORMALIZE, opts[1])
            eng.set_option(OPT_FAST_KERNEL, opts[2])
            eng.set_path(path)
            eng.set_outputs(perbase=(mode == "full"), sums_only=(mode == "sums"))
            eng.set_params(window_size=W, min_mapq=Q, min_cov=mincov, max_mean_depth=maxmean, step=step)
            eng.set_contigs(lens)
            for t, r in reads.items():
                _split_push(eng, rng, t, r)
            n_compute = int(rng.integers(1, 3))
            if os.environ.get("GOLEFT_SOAK_VERBOSE"):
                print("soak", seed, job, "path", path, mode, "W", W, "Q", Q, "opts", opts, "lens", lens, "x%d" % n_compute,
                      {t: (len(r.pos), len(r.cigar)) for t, r in reads.items()}, flush=True)
            for _ in range(n_compute):                                   # a second pass over resident records
                eng.compute()
            eff_step = step if step else po.step_for(W)
            tag = (seed, job, path, mode, W, Q, mincov, maxmean, step, lens)
            for t, L in enumerate(lens):
                want = po.perbase_c(reads.get(t, H.empty_reads()), Q, 0, L) if L else np.zeros(0, np.int32)
                ws, wm = H.oracle_windows(want, W)
                if mode == "sums":
                    assert np.array_equal(eng.window_sums(t), ws), tag + (t,)    # (minima / runs: only when the
                    continue                                                       # regular kernel had to run)
                gs, gm = eng.windows(t)
                assert np.array_equal(gs, ws), tag + (t, "sums")
                assert np.array_equal(gm, wm), tag + (t, "mins")
                assert np.array_equal(eng.callable_runs(t), H.oracle_runs(want, mincov, maxmean, eff_step)), tag + (t, "runs")
                if mode == "full":
                    got = eng.perbase(t)
                    assert np.array_equal(got, want), tag + (t, "perbase")
                    if L > 2 and rng.random() < 0.5:
                        a = int(rng.integers(0, L)); b = int(rng.integers(a, L + 1))
                        s1, m1 = eng.region_windows(t, a, b)
                        rs, rm = H.oracle_windows(want[a:b], W, a)
                        assert np.array_equal(s1, rs) and np.array_equal(m1, rm), tag + (t, a, b)
                elif L:
                    with pytest.raises(GdError):
                        eng.perbase(t)
