"""-m gpu: gd_compute in two halves (gd_compute_launch / gd_compute_finish) -- what bench.py's multi-GPU loop uses
to issue the previous step's collective while this step's kernels run.  Same results as gd_compute, including the
attempts that finish() has to repeat (a read longer than the look-back; AUTO leaving the tile path), and the call
order is enforced."""
import numpy as np
import pytest

from oracle import pyoracle as po
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _results(eng, lengths):
    out = []
    for t in range(len(lengths)):
        s, m = eng.windows(t)
        out.append((eng.perbase(t), s, m, eng.callable_runs(t)))
    return out


@pytest.mark.parametrize("path", [0, 1, 3])                   # AUTO, TILE, CHUNK
def test_launch_finish_equals_compute(path):
    from goleft_amd.engine import DepthEngine
    rng = np.random.default_rng(11 + path)
    lengths = [150_000, 4096, 1, 70_001]
    reads = {0: H.random_reads(rng, lengths[0], 20000, max_len=150),
             1: H.random_reads(rng, lengths[1], 500, max_len=90),
             3: H.random_reads(rng, lengths[3], 3000, max_len=300, long_reads=(path != 1))}
    res = []
    for split in (False, True):
        with DepthEngine(0) as eng:
            eng.set_params(window_size=100, min_mapq=1, min_cov=4)
            eng.set_path(path)
            eng.set_contigs(lengths)
            for t, r in reads.items():
                eng.push(t, r.pos, r.flag, r.mapq, r.cigar_off, r.cigar)
            for _ in range(3):                                 # several steps: the second launch reuses every buffer
                if split:
                    eng.compute_launch()
                    eng.compute_finish()
                else:
                    eng.compute()
            res.append((_results(eng, lengths), eng.stats().reruns, eng.stats().path))
    for a, b in zip(res[0][0], res[1][0]):
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
    assert res[0][1:] == res[1][1:]
    for t, L in enumerate(lengths):
        want = po.perbase_c(reads.get(t, H.empty_reads()), 1, 0, L)
        assert np.array_equal(res[1][0][t][0], want)


def test_finish_repeats_a_failed_attempt_and_the_order_is_enforced():
    from goleft_amd.engine import DepthEngine, GdError
    L = 50_000
    # one read of 1500 bases: longer than the default look-back of 512 -> the first attempt is repeated inside finish()
    pos = np.array([100, 200, 30000], np.int32)
    cig = np.array([(50 << 4), (1500 << 4), (80 << 4)], np.uint32)
    r = po.Reads(pos, np.zeros(3, np.uint16), np.full(3, 60, np.uint8), np.arange(4, dtype=np.uint32), cig)
    with DepthEngine(0) as eng:
        eng.set_params(window_size=250, min_mapq=1, min_cov=1)
        eng.set_path(1)
        eng.set_option(14, 0)                                  # GD_OPT_INGEST_INDEX off: spans are not measured at arrival, the
        eng.set_contigs([L])                                   # look-back starts at the default and has to be learnt
        eng.push(0, r.pos, r.flag, r.mapq, r.cigar_off, r.cigar)
        with pytest.raises(GdError):
            eng.compute_finish()                               # nothing launched
        eng.compute_launch()
        with pytest.raises(GdError):
            eng.compute_launch()                               # one in flight
        with pytest.raises(GdError):
            eng.set_params(window_size=100)                    # the job may not change under it
        with pytest.raises(GdError):
            eng.push(0, r.pos, r.flag, r.mapq, r.cigar_off, r.cigar)
        eng.compute_finish()
        st = eng.stats()
        assert st.reruns == 1 and st.max_span_seen == 1500
        assert np.array_equal(eng.perbase(0), po.perbase_c(r, 1, 0, L))
        eng.compute()                                          # and the synchronous form still works afterwards
        assert eng.stats().reruns == 0


def test_producers_fill_the_ring_in_place_and_push_agree():
    """Two ways from host memory into HBM -- gd_push (library threads copy and validate) and the public-ABI producer of
    the host library (gd_reserve, gd_acquire, N threads write the block in place, gd_commit validates) -- with blocks
    small enough that both ring and worker pool turn over many times: same records, same depth; an unsorted block is
    refused by either with GD_E_UNSORTED and the context goes on."""
    import numpy as np
    from goleft_amd import _hostlib, synth
    from goleft_amd.engine import DepthEngine, GdError, OPT_PUSH_CHUNK, OPT_PUSH_THREADS
    from oracle import pyoracle as po
    L = 4_000_000
    r = po.Reads(*synth.short_reads_numpy(L, synth.n_reads_for(L), 9))
    want = po.perbase_c(r, 1, 0, L)
    host = _hostlib.load()
    with DepthEngine(0) as eng:
        eng.set_params(window_size=1000)
        eng.set_contigs([L, L])
        eng.set_option(OPT_PUSH_CHUNK, 1 << 18)
        eng.set_option(OPT_PUSH_THREADS, 5)
        eng.push(0, r.pos, r.flag, r.mapq, r.cigar_off, r.cigar)
        rc = host.gdh_produce_in_place(eng._ctx, 1, r.pos.ctypes.data, r.flag.ctypes.data, r.mapq.ctypes.data,
                                       r.cigar_off.ctypes.data, r.cigar.ctypes.data, r.n, r.n_ops, 7, 1 << 18)
        assert rc == 0
        eng.compute()
        assert np.array_equal(eng.perbase(0), want) and np.array_equal(eng.perbase(1), want)
        bad = r.pos.copy()
        bad[600_000], bad[600_001] = bad[600_001] + 5, bad[600_000]          # out of order inside a later block
        eng.reset()
        with pytest.raises(GdError) as ei:
            eng.push(0, bad, r.flag, r.mapq, r.cigar_off, r.cigar)
        assert ei.value.status == -7
        eng.reset()
        rc = host.gdh_produce_in_place(eng._ctx, 1, bad.ctypes.data, r.flag.ctypes.data, r.mapq.ctypes.data,
                                       r.cigar_off.ctypes.data, r.cigar.ctypes.data, r.n, r.n_ops, 7, 1 << 18)
        assert rc == -7
        eng.reset()
        eng.push(0, r.pos, r.flag, r.mapq, r.cigar_off, r.cigar)
        eng.compute()
        assert np.array_equal(eng.perbase(0), want)


def test_csr_offsets_that_dip_exactly_at_a_work_item_cut_are_refused():
    """gd_push's filler threads cut the CSR offsets into 1 MB work items (262 144 offsets) without
    overlap; a decrease exactly at a cut -- even below the block's first offset, so that the rebased value wraps --
    must be refused like any other (GD_E_INVALID), by gd_push and by the in-place producer, and nothing may be
    written outside the pinned block on the way."""
    import numpy as np
    from goleft_amd import _hostlib, synth
    from goleft_amd.engine import DepthEngine, GdError
    from oracle import pyoracle as po
    L = 3_000_000
    r = po.Reads(*synth.short_reads_numpy(L, synth.n_reads_for(L), 11))
    assert r.n > 2 * 262144 + 10, r.n
    host = _hostlib.load()
    with DepthEngine(0) as eng:
        eng.set_params(window_size=1000)
        eng.set_contigs([L, L])
        pos, flag, mapq = (np.ascontiguousarray(a[5:]) for a in (r.pos, r.flag, r.mapq))   # (block offsets start above 0)
        assert r.cigar_off[5] >= 5
        for cut, low in ((262144, 3), (2 * 262144, 0), (262144, None)):
            bad = r.cigar_off[5:].copy()
            bad[cut] = bad[cut - 1] - 1 if low is None else low     # below its predecessor; `low` is below the block's start too
            eng.reset()
            with pytest.raises(GdError) as ei:
                eng.push(0, pos, flag, mapq, bad, r.cigar)
            assert ei.value.status == -1, ei.value.status
            eng.reset()
            rc = host.gdh_produce_in_place(eng._ctx, 1, pos.ctypes.data, flag.ctypes.data, mapq.ctypes.data,
                                           bad.ctypes.data, r.cigar.ctypes.data, len(pos), r.n_ops, 7, 1 << 20)
            assert rc == -1, rc
        eng.reset()
        eng.push(0, r.pos, r.flag, r.mapq, r.cigar_off, r.cigar)
        eng.compute()
        assert np.array_equal(eng.perbase(0), po.perbase_c(r, 1, 0, L))
