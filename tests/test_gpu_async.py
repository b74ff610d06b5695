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


def test_a_producer_holds_three_blocks_and_the_fourth_is_refused():
    """gd_acquire hands out the next block before the last one is committed (include/goleft_depth.h: up to three held,
    so a decoder's threads write block k+1 while gd_commit validates and sends block k): three blocks filled up front
    and committed afterwards give the depth of one gd_push; the acquire that comes round to a held block, and the
    second commit of one block, are GD_E_STATE; n_reads 0 and gd_reset give blocks back."""
    import ctypes as C
    import numpy as np
    from goleft_amd import synth
    from goleft_amd.engine import DepthEngine, GdError
    from oracle import pyoracle as po
    L = 600_000
    r = po.Reads(*synth.short_reads_numpy(L, synth.n_reads_for(L), 5))
    want = po.perbase_c(r, 1, 0, L)

    def fill(b, a, e):
        n, o0, o1 = e - a, int(r.cigar_off[a]), int(r.cigar_off[e])
        for name, src, ct in (("pos", r.pos[a:e], C.c_int32), ("flag", r.flag[a:e], C.c_uint16), ("mapq", r.mapq[a:e], C.c_uint8),
                              ("cigar_off", r.cigar_off[a:e + 1] - r.cigar_off[a], C.c_uint32), ("cigar", r.cigar[o0:o1], C.c_uint32)):
            src = np.ascontiguousarray(src)
            C.memmove(getattr(b, name), src.ctypes.data, src.nbytes)
        return n, o1 - o0

    cuts = [0, r.n // 3, 2 * r.n // 3, r.n]
    with DepthEngine(0) as eng:
        eng.set_params(window_size=1000)
        eng.set_contigs([L])
        held = []
        for a, e in zip(cuts[:-1], cuts[1:]):
            b = eng.acquire(e - a, int(r.cigar_off[e] - r.cigar_off[a]))
            held.append((b, fill(b, a, e)))
        assert len({b.slot for b, _ in held}) == 3
        spare = eng.acquire(16, 16)                          # the fourth slot
        with pytest.raises(GdError) as ei:
            eng.acquire(16, 16)                              # comes round to the first, still held
        assert ei.value.status == -4, ei.value.status
        eng.commit(spare, 0, 0, 0)                           # given back unused
        for b, (n, m) in held:
            eng.commit(b, 0, n, m)
        with pytest.raises(GdError) as ei:
            eng.commit(held[0][0], 0, *held[0][1])
        assert ei.value.status == -4, ei.value.status
        eng.compute()
        assert np.array_equal(eng.perbase(0), want)
        # blocks held across a reset are the ring's again
        for _ in range(3):
            eng.acquire(16, 16)
        eng.reset()
        for _ in range(2):
            for a, e in zip(cuts[:-1], cuts[1:]):
                b = eng.acquire(e - a, int(r.cigar_off[e] - r.cigar_off[a]))
                eng.commit(b, 0, *fill(b, a, e))
            eng.compute()
            assert np.array_equal(eng.perbase(0), want)
            eng.reset()


def test_commits_checked_on_the_device_give_the_host_checks_answers_later():
    """GD_OPT_COMMIT_CHECK = 1: gd_commit leaves a block's checks (coordinate order, negative positions, CSR offsets) to
    the pass that indexes it on the device; gd_check_commits -- or the next gd_compute -- gives the answer the host
    check gives inside gd_commit, for a fault in the middle of a block, at a block's first record (the seam: at once)
    and at its last, and keeps giving it until gd_reset; good records give the depth of a gd_push; the option reads
    back, and gdh_produce_in_place leaves it as it found it."""
    import ctypes as C
    import numpy as np
    from goleft_amd import _hostlib, synth
    from goleft_amd.engine import DepthEngine, GdError, OPT_COMMIT_CHECK
    from oracle import pyoracle as po
    L = 400_000
    r = po.Reads(*synth.short_reads_numpy(L, synth.n_reads_for(L), 3))
    want = po.perbase_c(r, 1, 0, L)
    cuts = [0, r.n // 2, r.n]
    assert r.n // 2 >= 4096

    def feed(eng, pos, off, tid=0):
        for a, e in zip(cuts[:-1], cuts[1:]):
            o0, o1 = int(off[a]), int(r.cigar_off[e])
            b = eng.acquire(e - a, o1 - int(r.cigar_off[a]))
            for name, src in (("pos", pos[a:e]), ("flag", r.flag[a:e]), ("mapq", r.mapq[a:e]),
                              ("cigar_off", off[a:e + 1] - r.cigar_off[a]), ("cigar", r.cigar[int(r.cigar_off[a]):o1])):
                src = np.ascontiguousarray(src)
                C.memmove(getattr(b, name), src.ctypes.data, src.nbytes)
            eng.commit(b, tid, e - a, o1 - int(r.cigar_off[a]))

    with DepthEngine(0) as eng:
        eng.set_params(window_size=1000)
        eng.set_contigs([L, L])
        assert eng.get_option(OPT_COMMIT_CHECK) == 0
        eng.set_option(OPT_COMMIT_CHECK, 1)
        assert eng.get_option(OPT_COMMIT_CHECK) == 1
        eng.check_commits()                                  # nothing committed: nothing to say
        feed(eng, r.pos, r.cigar_off)
        eng.check_commits()
        eng.compute()
        assert np.array_equal(eng.perbase(0), want)
        mid = cuts[1] + 1000
        for what, status in (("order", -7), ("negative", -5), ("offsets", -1), ("last", -7)):
            pos, off = r.pos.copy(), r.cigar_off.copy()
            if what == "order":
                pos[mid] = pos[mid - 1] - 1
            elif what == "negative":
                pos[0] = 0
                pos[1:3] = -1                                # (out of order as well: the negative position is what is reported)
            elif what == "offsets":
                off[mid] = off[mid - 1] - 1
            else:
                pos[r.n - 1] = pos[r.n - 2] - 1
            eng.reset()
            feed(eng, pos, off, tid=1)                       # gd_commit has nothing to say yet
            for _ in range(2):
                with pytest.raises(GdError) as ei:
                    eng.check_commits()
                assert ei.value.status == status, (what, ei.value.status)
            with pytest.raises(GdError):
                eng.compute()
        # the seam with the records in front is the host's business even then
        eng.reset()
        pos = r.pos.copy()
        pos[cuts[1]] = pos[cuts[1] - 1] - 1
        with pytest.raises(GdError) as ei:
            feed(eng, pos, r.cigar_off)
        assert ei.value.status == -7
        eng.reset()
        eng.set_option(OPT_COMMIT_CHECK, 0)
        host = _hostlib.load()
        rc = host.gdh_produce_in_place(eng._ctx, 0, r.pos.ctypes.data, r.flag.ctypes.data, r.mapq.ctypes.data,
                                       r.cigar_off.ctypes.data, r.cigar.ctypes.data, r.n, r.n_ops, 4, 8192)
        assert rc == 0 and eng.get_option(OPT_COMMIT_CHECK) == 0
        eng.compute()
        assert np.array_equal(eng.perbase(0), want)
