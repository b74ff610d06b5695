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
# GOLEFT_SOAK_ONLY="13,14": run only these jobs of each seed (every job draws from its own generator)
ONLY = [int(x) for x in os.environ.get("GOLEFT_SOAK_ONLY", "").split(",") if x]


def _split_push(eng, rng, tid, r):
    """Push a contig's records in 1..3 consecutive batches."""
    n = len(r.pos)
    cuts = sorted(set([0, n] + [int(x) for x in rng.integers(0, n + 1, size=int(rng.integers(0, 3)))]))
    for a, b in zip(cuts[:-1], cuts[1:]):
        if a == b:
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



def _adopt(eng, torch, tid, r):
    dev = torch.device("cuda", 0)
    eng.adopt_device(tid, torch.from_numpy(r.pos).to(dev), torch.from_numpy(r.flag.view(np.int16)).to(dev),
                     torch.from_numpy(r.mapq).to(dev), torch.from_numpy(r.cigar_off.view(np.int32)).to(dev),
                     torch.from_numpy(r.cigar.view(np.int32)).to(dev))


def _concat(a, b):
    return po.Reads(np.concatenate([a.pos, b.pos]), np.concatenate([a.flag, b.flag]), np.concatenate([a.mapq, b.mapq]),
                    np.concatenate([a.cigar_off, b.cigar_off[1:] + a.cigar_off[-1]]).astype(np.uint32),
                    np.concatenate([a.cigar, b.cigar]))


@pytest.mark.parametrize("seed", range(int(os.environ.get("GOLEFT_SOAK_SEEDS", "4"))))
def test_soak_one_context_many_jobs(seed):
    import torch
    from goleft_amd.engine import (DepthEngine, GdError, PATH_AUTO, PATH_TILE, PATH_SCATTER, PATH_CHUNK,
                                   OPT_NT_STORES, OPT_FAST_KERNEL)
    with DepthEngine(0) as eng:
        for job in range(JOBS):
            if ONLY and job not in ONLY:
                continue
            rng = np.random.default_rng([9000 + seed, job])
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
            adopt = bool(rng.random() < 0.3)
            split = bool(rng.random() < 0.3)
            eng.set_option(OPT_NT_STORES, opts[0])
            eng.set_option(OPT_FAST_KERNEL, opts[2])
            eng.set_path(path)
            eng.set_outputs(perbase=(mode == "full"), sums_only=(mode == "sums"))
            eng.set_params(window_size=W, min_mapq=Q, min_cov=mincov, max_mean_depth=maxmean, step=step)
            eng.set_contigs(lens)
            for t, r in reads.items():
                if adopt:
                    _adopt(eng, torch, t, r)
                else:
                    _split_push(eng, rng, t, r)
            selected = list(range(n_ctg))
            if n_ctg > 1 and rng.random() < 0.25:            # a job over some of the contigs
                selected = sorted(int(x) for x in rng.choice(n_ctg, size=int(rng.integers(1, n_ctg)), replace=False))
                eng.select_contigs(selected)
            n_compute = int(rng.integers(1, 4))
            if os.environ.get("GOLEFT_SOAK_VERBOSE"):
                print("soak", seed, job, "path", path, mode, "W", W, "Q", Q, "opts", opts, "adopt", adopt, "split", split,
                      "lens", lens, "sel", selected, "x%d" % n_compute,
                      {t: (len(r.pos), len(r.cigar)) for t, r in reads.items()}, flush=True)
            for k in range(n_compute):
                if k and not adopt and rng.random() < 0.5:   # more records arrive between two computes
                    t = int(rng.integers(0, n_ctg))
                    last = int(reads[t].pos[-1]) if t in reads and len(reads[t].pos) else 0
                    if lens[t] - last > 1:
                        more = H.random_reads(rng, lens[t] - last, int(rng.integers(1, 800)),
                                              long_reads=bool(rng.random() < 0.3))
                        more = po.Reads((more.pos + last).astype(np.int32), more.flag, more.mapq, more.cigar_off, more.cigar)
                        eng.push(t, more.pos, more.flag, more.mapq, more.cigar_off, more.cigar)
                        reads[t] = _concat(reads[t], more) if t in reads else more
                if split:
                    eng.compute_launch()
                    eng.compute_finish()
                else:
                    eng.compute()
            eff_step = step if step else po.step_for(W)
            tag = (seed, job, path, mode, W, Q, mincov, maxmean, step, lens)
            per = {}
            for t in selected:
                L = lens[t]
                want = po.perbase_c(reads.get(t, H.empty_reads()), Q, 0, L) if L else np.zeros(0, np.int32)
                per[t] = want
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
                elif L:
                    with pytest.raises(GdError):
                        eng.perbase(t)
            if mode == "full" and rng.random() < 0.5:        # the --bed reductions, many regions in one call
                rt, rs, re_ = [], [], []
                for _ in range(int(rng.integers(1, 60))):
                    t = int(rng.choice(selected))
                    a = int(rng.integers(0, max(1, lens[t]) + 20))
                    rt.append(t); rs.append(a); re_.append(a + int(rng.choice([0, 1, 9, 300, 5000, 400000])))
                sums, mins, runs = eng.regions(rt, rs, re_)
                for k, (t, a, b) in enumerate(zip(rt, rs, re_)):
                    if b == a:
                        assert len(sums[k]) == 0 and len(runs[k]) == 0
                        continue
                    d = np.zeros(b - a, np.int32)
                    hi = min(b, lens[t])
                    if hi > a:
                        d[:hi - a] = per[t][a:hi]
                    ws, wm = H.oracle_windows(d, W, a)
                    assert np.array_equal(sums[k], ws), tag + ("region", t, a, b)
                    assert np.array_equal(mins[k], wm if lens[t] else np.zeros(len(wm), np.int32)), tag + ("region", t, a, b)
                    assert np.array_equal(runs[k], H.oracle_runs(d, mincov, maxmean, 1 << 62, a)), tag + ("region", t, a, b)
            if len(selected) != n_ctg:
                eng.select_contigs([])
