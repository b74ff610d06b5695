"""GPU: the position index and the span measurement that gd_index_records_kernel leaves as records arrive
(GD_OPT_INGEST_INDEX, default on): whatever way the records came -- adopted device arrays,
blocks committed through the ring, several pushes -- gd_prep_kernel looks its tiles' read ranges up in that index
instead of searching, and the FIRST gd_compute already runs with the look-back the data needs.  Every case is
compared with the C oracle (oracle/depth_oracle.c restating `samtools depth` as depth/depth.go:45 calls it) and
with the engine run with the index switched off."""
import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu


def _gappy_reads(rng, L, n, gaps):
    """short reads whose start positions avoid `gaps` [(a, b)]: the index entries of a gap belong to the first read
    behind it -- tens of thousands of them for one read (a centromere)"""
    from oracle import pyoracle as po
    r = H.random_reads(rng, L, n, max_ops=4, max_len=120)
    pos = r.pos.astype(np.int64)
    for a, b in gaps:
        inside = (pos >= a) & (pos < b)
        pos[inside] = b + (pos[inside] - a) % 977
    order = np.argsort(pos, kind="stable")
    nops = np.diff(r.cigar_off.astype(np.int64))[order]
    off = np.zeros(n + 1, np.uint32)
    off[1:] = np.cumsum(nops)
    cig = np.concatenate([r.cigar[r.cigar_off[i]:r.cigar_off[i + 1]] for i in order]) if n else r.cigar
    return po.Reads(pos[order].astype(np.int32), r.flag[order], r.mapq[order], off, cig.astype(np.uint32))


def _results(eng, tids):
    out = []
    for t in tids:
        s, m = eng.windows(t)
        out.append((eng.perbase(t), s, m, eng.callable_runs(t)))
    return out


@pytest.mark.parametrize("how", ["adopt", "push", "push-in-blocks", "commit-small-blocks"])
def test_index_built_on_arrival_equals_search_and_oracle(how):
    import torch
    from goleft_amd.engine import DepthEngine, OPT_INGEST_INDEX, OPT_PUSH_CHUNK
    from oracle import pyoracle as po
    rng = np.random.default_rng(41)
    lens = [3_000_000, 70_000, 4096, 1, 900_001]
    reads = [_gappy_reads(rng, lens[0], 60_000, [(200_000, 1_500_000), (2_000_000, 2_000_100)]),
             _gappy_reads(rng, lens[1], 9_000, []),
             H.random_reads(rng, lens[2], 300, max_ops=3, max_len=80),
             H.empty_reads(),
             _gappy_reads(rng, lens[4], 5_000, [(0, 600_000)])]        # nothing in front of the first read
    got = {}
    for index in (1, 0):
        with DepthEngine(0) as eng:
            eng.set_params(window_size=1000, min_mapq=1, min_cov=4)
            eng.set_option(OPT_INGEST_INDEX, index)
            eng.set_contigs(lens)
            keep = []
            for t, r in enumerate(reads):
                if how == "adopt":
                    if r.n == 0:
                        continue
                    dev = [torch.from_numpy(np.ascontiguousarray(a).view(v)).cuda() for a, v in
                           ((r.pos, np.int32), (r.flag, np.int16), (r.mapq, np.uint8), (r.cigar_off, np.int32),
                            (r.cigar if r.n_ops else np.zeros(1, np.uint32), np.int32))]
                    keep.append(dev)
                    eng.adopt_device(t, dev[0], dev[1], dev[2], dev[3], dev[4][:r.n_ops])
                elif how == "push":
                    eng.push(t, r.pos, r.flag, r.mapq, r.cigar_off, r.cigar)
                else:
                    if how == "commit-small-blocks":
                        eng.set_option(OPT_PUSH_CHUNK, 4096)
                    cuts = sorted(set([0, r.n] + [int(x) for x in rng.integers(0, r.n + 1, size=5)])) if r.n else [0, 0]
                    for a, b in zip(cuts[:-1], cuts[1:]):                # several pushes of one contig
                        if b > a:
                            o = r.cigar_off[a:b + 1]
                            eng.push(t, r.pos[a:b], r.flag[a:b], r.mapq[a:b], o - o[0], r.cigar[o[0]:o[-1]])
            eng.compute()
            st = eng.stats()
            assert st.reruns == 0
            if index:
                # spans were measured as the records arrived: the look-back is the data's, not the default 512
                true_span = max(int(H.ref_span(r).max()) if r.n else 0 for r in reads)
                assert st.lookback >= st.max_span_seen and st.lookback == max(64, (true_span + 63) & ~63), (st.lookback, true_span)
            got[index] = _results(eng, range(len(lens)))
    for t, r in enumerate(reads):
        want = po.perbase_c(r, 1, 0, lens[t])
        for index in (1, 0):
            d, s, m, runs = got[index][t]
            assert np.array_equal(d, want), (how, index, t)
            ws, wm = H.oracle_windows(want, 1000)
            assert np.array_equal(s, ws) and np.array_equal(m, wm)
        assert np.array_equal(got[1][t][3], got[0][t][3])


def test_first_compute_needs_no_second_attempt_and_a_long_span_still_reruns():
    """The measured span is over ALL records (unfiltered) and only of reads of at most 64 ops: a kept read the pass
    did not walk must still be caught by the tile kernel's verification (one re-run), bit exact either way."""
    from goleft_amd.engine import DepthEngine
    from oracle import pyoracle as po
    rng = np.random.default_rng(5)
    L = 500_000
    r = H.random_reads(rng, L, 20_000, max_ops=3, max_len=60)
    # one read of 70 ops (not walked at arrival) spanning ~7 kb
    k = 10_000
    nops = np.diff(r.cigar_off.astype(np.int64))
    nops[k] = 70
    off = np.zeros(r.n + 1, np.uint32)
    off[1:] = np.cumsum(nops)
    cig = np.zeros(int(off[-1]), np.uint32)
    for i in range(r.n):
        a, b = int(off[i]), int(off[i + 1])
        if i == k:
            cig[a:b] = (100 << 4) | 0
        else:
            src = r.cigar[r.cigar_off[i]:r.cigar_off[i + 1]]
            cig[a:b] = src
    flag = r.flag.copy(); flag[k] = 0
    mapq = r.mapq.copy(); mapq[k] = 60
    r2 = po.Reads(r.pos, flag, mapq, off, cig)
    with DepthEngine(0) as eng:
        eng.set_params(window_size=250)
        eng.set_contigs([L])
        eng.push(0, r2.pos, r2.flag, r2.mapq, r2.cigar_off, r2.cigar)
        eng.compute()
        st = eng.stats()
        assert st.reruns == 1 and st.max_span_seen == 7000
        assert np.array_equal(eng.perbase(0), po.perbase_c(r2, 1, 0, L))
        eng.compute()
        assert eng.stats().reruns == 0
