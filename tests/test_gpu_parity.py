"""-m gpu: HIP engine (through the C ABI) vs the CPU oracle, bit exact."""
import numpy as np
import pytest

from tests import helpers as H
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from goleft_amd.engine import DepthEngine
    e = DepthEngine(0)
    yield e
    e.close()


def run_engine(eng, contigs, reads, **params):
    eng.set_params(**params)
    eng.set_contigs([c[1] for c in contigs])
    for tid, r in reads.items():
        eng.push(tid, r.pos, r.flag, r.mapq, r.cigar_off, r.cigar)
    eng.compute()


def check_all(eng, contigs, reads, W, Q, mincov, maxmean, step=None):
    if step is None:
        step = po.step_for(W)
    for tid, (_, clen) in enumerate(contigs):
        r = reads.get(tid, H.empty_reads())
        want = po.perbase_c(r, Q, 0, clen)
        got = eng.perbase(tid)
        assert np.array_equal(got, want), "per-base mismatch tid %d: first at %d" % (
            tid, int(np.nonzero(got != want)[0][0]))
        ws, wm = H.oracle_windows(want, W)
        gs, gm = eng.windows(tid)
        assert np.array_equal(gs, ws), "window sums tid %d" % tid
        assert np.array_equal(gm, wm), "window mins tid %d" % tid
        wr = H.oracle_runs(want, mincov, maxmean, step)
        gr = eng.callable_runs(tid)
        assert np.array_equal(gr, wr), "callable runs tid %d" % tid


@pytest.mark.parametrize("name", ["t", "hla", "t_empty"])
@pytest.mark.parametrize("W,Q,mincov,maxmean", [(1000, 1, 4, 0), (250, 1, 4, 0), (13, 0, 10, 1500),
                                                (1000000000, 1, 4, 0)])
def test_fixture_streams(eng, name, W, Q, mincov, maxmean):
    contigs, reads, z = H.load_golden_bam(name)
    run_engine(eng, contigs, reads, window_size=W, min_mapq=Q, min_cov=mincov, max_mean_depth=maxmean)
    check_all(eng, contigs, reads, W, Q, mincov, maxmean)
    if Q == 1:
        for tid in reads:
            assert np.array_equal(eng.perbase(tid), z["perbase_Q1_%d" % tid])


@pytest.mark.parametrize("seed", range(6))
def test_random_cigars(eng, seed):
    rng = np.random.default_rng(seed)
    lens = [int(rng.integers(1, 60000)) for _ in range(4)] + [1, 8192, 8193, 16384]
    contigs = [("c%d" % i, l) for i, l in enumerate(lens)]
    reads = {}
    for tid, l in enumerate(lens):
        if tid == 2:
            continue  # a contig without records
        reads[tid] = H.random_reads(rng, l, int(rng.integers(0, 4000)), long_reads=(seed % 2 == 1))
    W = int(rng.choice([1, 3, 7, 100, 250, 1000, 5000]))
    step = W * int(rng.integers(1, 50))
    mincov = int(rng.integers(1, 8))
    maxmean = int(rng.choice([0, 20]))
    run_engine(eng, contigs, reads, window_size=W, min_mapq=1, min_cov=mincov,
               max_mean_depth=maxmean, step=step)
    check_all(eng, contigs, reads, W, 1, mincov, maxmean, step=step)


def test_synthetic_short_reads(eng):
    from goleft_amd import synth
    L = 3_000_000
    n = synth.n_reads_for(L)
    r = po.Reads(*synth.short_reads_numpy(L, n, 20))
    contigs = [("chrS", L)]
    run_engine(eng, contigs, {0: r}, window_size=1000, min_mapq=1, min_cov=4)
    check_all(eng, contigs, {0: r}, 1000, 1, 4, 0)
    st = eng.stats()
    assert st.n_reads == n and st.reruns == 0
