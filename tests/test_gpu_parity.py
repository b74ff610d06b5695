"""-m gpu: HIP engine (through the C ABI) vs the CPU oracle, bit exact."""
import numpy as np
import pytest

from tests import helpers as H
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


PATH_IDS = {"tile": 1, "scatter": 2, "chunk": 3}


@pytest.fixture(scope="module", params=["tile", "chunk", "scatter"])
def eng(request):
    """Every parity test runs on all three device algorithms, each on the records as they arrived (GD_PATH_TILE: LDS
    tiles re-walking whole CIGARs, the short-read path; GD_PATH_CHUNK: LDS tiles over deletion lists, the long-read
    path; GD_PATH_SCATTER: global scatter + in-place scan)."""
    from goleft_amd.engine import DepthEngine
    e = DepthEngine(0)
    e.set_path(PATH_IDS[request.param])
    e.path_name = request.param
    yield e
    e.close()


@pytest.fixture(scope="module")
def auto_eng():
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
    assert st.path == PATH_IDS[eng.path_name]


def _uniform_reads(pos, length, flag=0, mapq=60):
    """n reads `<length>M` at the given sorted positions."""
    n = len(pos)
    off = np.arange(n + 1, dtype=np.uint32)
    cigar = np.full(n, (length << 4) | 0, np.uint32)
    return po.Reads(np.asarray(pos, np.int32), np.full(n, flag, np.uint16), np.full(n, mapq, np.uint8),
                    off, cigar)


def test_wide_pileup_64bit_windows(eng):
    """More than 2^22 reads on one tile: depth may exceed what the 32-bit window
    accumulation holds, so the tile takes the 64-bit path (and many record
    batches, CIGARs read from global memory)."""
    rng = np.random.default_rng(5)
    n = (1 << 22) + 70001
    L = 9000
    pos = np.sort(rng.integers(100, 140, size=n)).astype(np.int32)
    r = _uniform_reads(pos, 120)
    contigs = [("deep", L)]
    run_engine(eng, contigs, {0: r}, window_size=1000, min_mapq=1, min_cov=4, max_mean_depth=3000000)
    check_all(eng, contigs, {0: r}, 1000, 1, 4, 3000000)
    assert int(eng.perbase(0).max()) > (1 << 22)


def test_many_batches_staged_boundary(eng):
    """Tiles holding a few thousand reads: several record batches per tile, with
    the CIGAR range on either side of the LDS staging capacity."""
    rng = np.random.default_rng(11)
    L = 40000
    for n in (1400, 1536, 1537, 5000, 20000):
        pos = np.sort(rng.integers(0, 4096 - 100, size=n)).astype(np.int32)
        r = _uniform_reads(pos, 100)
        # sprinkle two-op reads (soft clip + match) so the queue path is used too
        extra = H.random_reads(rng, L, 3000)
        contigs = [("a", L), ("b", L)]
        run_engine(eng, contigs, {0: r, 1: extra}, window_size=250, min_mapq=1, min_cov=4)
        check_all(eng, contigs, {0: r, 1: extra}, 250, 1, 4, 0)


@pytest.mark.parametrize("index", [1, 0])
def test_lookback_adapts_and_stays_exact(eng, index):
    """index = 1 (GD_OPT_INGEST_INDEX, the default): the largest span is measured as the records arrive, so every
    compute -- the first one, and the one after longer reads were appended -- runs with the right look-back at once.
    index = 0: the look-back starts at the default, shrinks after a compute whose longest read is far below it, grows
    again (one re-run) when longer reads arrive.  Results stay exact either way."""
    from goleft_amd.engine import OPT_INGEST_INDEX
    if not eng.path_name.startswith("tile"):
        pytest.skip("the look-back exists only on the tile path")
    eng.set_option(OPT_INGEST_INDEX, index)
    rng = np.random.default_rng(3)
    L = 200000
    short = _uniform_reads(np.sort(rng.integers(0, L // 2, size=30000)), 50)
    contigs = [("c", L)]
    run_engine(eng, contigs, {0: short}, window_size=1000, min_mapq=1, min_cov=4)
    check_all(eng, contigs, {0: short}, 1000, 1, 4, 0)
    st = eng.stats()
    assert st.max_span_seen == 50 and st.reruns == 0 and st.lookback == (64 if index else 512)
    eng.compute()                                   # same data again: tighter, still exact
    st = eng.stats()
    assert st.lookback == 64 and st.reruns == 0
    check_all(eng, contigs, {0: short}, 1000, 1, 4, 0)
    # append longer reads (coordinate order kept): the tightened look-back is now too small
    longer = _uniform_reads(np.sort(rng.integers(L // 2, L - 2000, size=5000)), 1500)
    eng.push(0, longer.pos, longer.flag, longer.mapq, longer.cigar_off, longer.cigar)
    eng.compute()
    st = eng.stats()
    assert st.reruns == (0 if index else 1) and st.max_span_seen == 1500 and st.lookback >= 1500
    both = po.Reads(np.concatenate([short.pos, longer.pos]), np.concatenate([short.flag, longer.flag]),
                    np.concatenate([short.mapq, longer.mapq]),
                    np.concatenate([short.cigar_off, longer.cigar_off[1:] + short.cigar_off[-1]]).astype(np.uint32),
                    np.concatenate([short.cigar, longer.cigar]))
    check_all(eng, contigs, {0: both}, 1000, 1, 4, 0)
    eng.set_option(OPT_INGEST_INDEX, 1)


def test_read_span_limit_is_an_error(eng):
    """Reference spans of 2^27 bases or more are rejected, not mis-counted."""
    from goleft_amd.engine import GdError
    L = 1 << 20
    off = np.array([0, 10], np.uint32)
    cigar = np.full(10, (((1 << 24) - 1) << 4) | 3, np.uint32)  # eight N skips of ~16.7 Mb between two M ops
    cigar[0] = (100 << 4) | 0                                   # (skips AFTER the last M would be dropped by the
    cigar[9] = (5 << 4) | 0                                     # normalisation: they cover nothing that is counted)
    r = po.Reads(np.array([10], np.int32), np.zeros(1, np.uint16), np.full(1, 60, np.uint8), off, cigar)
    eng.set_params(window_size=1000, min_mapq=1, min_cov=4)
    eng.set_contigs([L])
    eng.push(0, r.pos, r.flag, r.mapq, r.cigar_off, r.cigar)
    if eng.path_name.startswith("tile"):
        with pytest.raises(GdError) as ei:
            eng.compute()
        assert ei.value.status == -5      # GD_E_RANGE
    else:                                 # the chunk and scatter paths have no span limit
        eng.compute()
        check_all(eng, [("c", L)], {0: r}, 1000, 1, 4, 0)


def test_auto_path_switches_on_long_spans(auto_eng):
    """GD_PATH_AUTO: short reads run on the tile path; a read spanning more than
    32768 bases moves the data set to the chunk path (one re-run), CIGAR-heavy
    records start there."""
    eng = auto_eng
    rng = np.random.default_rng(17)
    L = 300000
    short = _uniform_reads(np.sort(rng.integers(0, L - 200, size=20000)), 150)
    contigs = [("c", L)]
    run_engine(eng, contigs, {0: short}, window_size=250, min_mapq=1, min_cov=4)
    assert eng.stats().path == 1
    check_all(eng, contigs, {0: short}, 250, 1, 4, 0)
    # one spliced read with a 60 kb N skip
    off = np.array([0, 3], np.uint32)
    cig = np.array([(50 << 4) | 0, (60000 << 4) | 3, (70 << 4) | 0], np.uint32)
    spliced = po.Reads(np.array([1000], np.int32), np.zeros(1, np.uint16), np.full(1, 60, np.uint8), off, cig)
    mixed = po.Reads(np.concatenate([spliced.pos, short.pos[short.pos >= 1000]]),
                     np.concatenate([spliced.flag, short.flag[short.pos >= 1000]]),
                     np.concatenate([spliced.mapq, short.mapq[short.pos >= 1000]]),
                     np.concatenate([[0], 3 + np.arange(0, (short.pos >= 1000).sum() + 1)]).astype(np.uint32),
                     np.concatenate([cig, short.cigar[short.pos >= 1000]]))
    run_engine(eng, contigs, {0: mixed}, window_size=250, min_mapq=1, min_cov=4)
    st = eng.stats()
    assert st.path == 3 and st.reruns == 1
    check_all(eng, contigs, {0: mixed}, 250, 1, 4, 0)
    # records averaging more than 6 ops go to the chunk path directly
    heavy = H.random_reads(rng, L, 5000, max_ops=40)
    run_engine(eng, contigs, {0: heavy}, window_size=250, min_mapq=1, min_cov=4)
    st = eng.stats()
    assert st.path == 3 and st.reruns == 0
    check_all(eng, contigs, {0: heavy}, 250, 1, 4, 0)


def test_synthetic_ont_long_reads(eng):
    """BASELINE.json config 5 in small: ~14 kb reads, ~1000 CIGAR ops each."""
    from goleft_amd import synth
    L = 1_500_000
    n = synth.n_ont_reads_for(L)
    r = po.Reads(*synth.ont_reads_numpy(L, n, 9))
    contigs = [("chrL", L), ("chrM2", 70000)]
    r2 = po.Reads(*synth.ont_reads_numpy(70000, 40, 10))
    run_engine(eng, contigs, {0: r, 1: r2}, window_size=1000, min_mapq=1, min_cov=4)
    check_all(eng, contigs, {0: r, 1: r2}, 1000, 1, 4, 0)


@pytest.mark.parametrize("seed", range(3))
def test_chunk_edges_and_huge_cigars(eng, seed):
    """CIGAR lengths on either side of the 64-op checkpoint chunk and of the 4096-op
    checkpoint block (strided probe), chunks that consume no reference, N skips that
    jump over whole tiles, reads hanging over the contig end."""
    rng = np.random.default_rng(100 + seed)
    L = [90000, 400000, 20000][seed]
    n_ops = [1, 2, 8, 9, 63, 64, 65, 127, 128, 129, 191, 192, 193, 1000, 4095, 4096, 4097, 8191, 8193,
             20000, 300, 64, 64, 128, 4160] * 3
    rng.shuffle(n_ops)
    r = H.long_cigar_reads(rng, L, n_ops, max_step=[12, 40, 3][seed], skip_every=[0, 5, 0][seed])
    # a read whose middle chunks consume no reference at all (insertions only)
    ins = np.concatenate([np.full(70, (5 << 4) | 0), np.full(200, (3 << 4) | 1), np.full(70, (7 << 4) | 0)]).astype(np.uint32)
    r2 = po.Reads(np.concatenate([r.pos, [r.pos[-1]]]).astype(np.int32), np.concatenate([r.flag, [0]]).astype(np.uint16),
                  np.concatenate([r.mapq, [60]]).astype(np.uint8),
                  np.concatenate([r.cigar_off, [r.cigar_off[-1] + len(ins)]]).astype(np.uint32),
                  np.concatenate([r.cigar, ins]).astype(np.uint32))
    contigs = [("lr", L), ("short", 5000)]
    few = H.long_cigar_reads(rng, 5000, [130, 70, 64], max_step=30)
    W = [1000, 250, 37][seed]
    run_engine(eng, contigs, {0: r2, 1: few}, window_size=W, min_mapq=1, min_cov=2)
    check_all(eng, contigs, {0: r2, 1: few}, W, 1, 2, 0)
    assert eng.stats().path == PATH_IDS[eng.path_name]


def test_chunk_windows_only_output():
    """The long-read tile path also runs without the per-base vector."""
    from goleft_amd import synth
    from goleft_amd.engine import DepthEngine, PATH_CHUNK
    L = 600_000
    r = po.Reads(*synth.ont_reads_numpy(L, synth.n_ont_reads_for(L), 12))
    with DepthEngine(0) as e:
        e.set_path(PATH_CHUNK)
        e.set_outputs(perbase=False)
        run_engine(e, [("c", L)], {0: r}, window_size=1000, min_mapq=1, min_cov=4)
        want = po.perbase_c(r, 1, 0, L)
        ws, wm = H.oracle_windows(want, 1000)
        gs, gm = e.windows(0)
        assert np.array_equal(gs, ws) and np.array_equal(gm, wm)
        assert np.array_equal(e.callable_runs(0), H.oracle_runs(want, 4, 0, po.step_for(1000)))
        assert e.stats().path == 3


def test_auto_path_picks_chunk_for_ont(auto_eng):
    from goleft_amd import synth
    L = 400_000
    r = po.Reads(*synth.ont_reads_numpy(L, synth.n_ont_reads_for(L), 4))
    run_engine(auto_eng, [("c", L)], {0: r}, window_size=250, min_mapq=1, min_cov=4)
    assert auto_eng.stats().path == 3 and auto_eng.stats().reruns == 0
    check_all(auto_eng, [("c", L)], {0: r}, 250, 1, 4, 0)


@pytest.mark.parametrize("W,size", [(250, 1000), (250, 600), (100, 100), (1000, 16000)])
def test_depthwed_matrix_on_device(auto_eng, W, size):
    """BASELINE.json config 4 in small: a cohort loaded as samples x contigs of one
    context; the device matrix equals what `goleft depthwed` prints from the samples'
    depth.bed files (oracle restatements of depth.go's callback and depthwed.go)."""
    from goleft_amd import synth
    eng = auto_eng
    ref = [("chrA", 100001), ("chrB", 35250), ("chrC", 999)]
    n_samples = 5
    rng = np.random.default_rng(21)
    contigs, reads, texts = [], {}, []
    for s in range(n_samples):
        per_sample = {}
        for j, (name, L) in enumerate(ref):
            cov = [3, 30, 60, 250, 0.4][s]
            n = int(L * cov / 150)
            r = po.Reads(*synth.short_reads_numpy(L, n, 100 * s + j)) if n else H.empty_reads()
            reads[len(contigs)] = r
            per_sample[j] = r
            contigs.append(("s%d.%s" % (s, name), L))
        texts.append(po.depth_run_oracle(ref, per_sample, W=W, Q=1, mincov=4)[0])
    run_engine(eng, contigs, reads, window_size=W, min_mapq=1, min_cov=4)
    tids = np.arange(n_samples * len(ref), dtype=np.int32).reshape(n_samples, len(ref))
    cells, ctg, st, en = eng.depthwed(tids, size)
    want = po.depthwed_py(texts, ["s%d" % s for s in range(n_samples)], size).strip().split("\n")[1:]
    assert len(want) == len(cells)
    for k, line in enumerate(want):
        t = line.split("\t")
        assert t[0] == ref[ctg[k]][0] and int(t[1]) == st[k] and int(t[2]) == en[k]
        assert [int(x) for x in t[3:]] == cells[k].tolist(), (k, line, cells[k])
    # the device-resident view holds the same matrix
    # (read back through the HIP runtime the engine already loaded; torch.cuda is kept out of
    # this process on purpose: it bundles its own copy of the runtime)
    import ctypes
    ptr, rows = eng.depthwed_device(tids, size)
    assert rows == len(cells)
    back = np.empty((rows, n_samples), np.int64)
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    assert hip.hipMemcpy(back.ctypes.data, ptr, back.nbytes, 2) == 0          # hipMemcpyDeviceToHost
    assert np.array_equal(back, cells)


def test_windows_only_output(auto_eng):
    """gd_set_outputs(0): no per-base vector in HBM; windows and class runs are unchanged,
    per-base consumers report GD_E_STATE."""
    from goleft_amd import synth
    from goleft_amd.engine import GdError
    eng = auto_eng
    L = 700_001
    r = po.Reads(*synth.short_reads_numpy(L, synth.n_reads_for(L), 33))
    contigs = [("c", L)]
    eng.set_outputs(perbase=False)
    try:
        run_engine(eng, contigs, {0: r}, window_size=250, min_mapq=1, min_cov=4)
        want = po.perbase_c(r, 1, 0, L)
        ws, wm = H.oracle_windows(want, 250)
        gs, gm = eng.windows(0)
        assert np.array_equal(gs, ws) and np.array_equal(gm, wm)
        assert np.array_equal(eng.callable_runs(0), H.oracle_runs(want, 4, 0, po.step_for(250)))
        with pytest.raises(GdError) as ei:
            eng.perbase(0)
        assert ei.value.status == -4
        with pytest.raises(GdError):
            eng.region_windows(0, 10, 1000)
    finally:
        eng.set_outputs(perbase=True)
    run_engine(eng, contigs, {0: r}, window_size=250, min_mapq=1, min_cov=4)
    assert np.array_equal(eng.perbase(0), want)


@pytest.mark.parametrize("stream", [1, 0])
@pytest.mark.parametrize("W", [32, 100, 250, 1000, 4096, 5000, 1 << 20])
def test_sums_only_output(W, stream):
    """gd_set_outputs(GD_OUT_SUMS_ONLY): window sums from read/window overlaps, no per-base scan -- the
    streaming kernel over the records as they arrived (gd_sums_stream.hpp; stream=1, the default) and the tile
    kernel it replaces (GD_OPT_FAST_KERNEL = 0); they equal the sums of the regular path; minima, class runs
    and the per-base vector report GD_E_STATE; the depthwed matrix is unchanged."""
    from goleft_amd import synth
    from goleft_amd import engine as E
    from goleft_amd.engine import DepthEngine, GdError, PATH_TILE, OPT_FAST_KERNEL
    rng = np.random.default_rng(W)
    lengths = [300_001, 1, 4096, 70_000, 12_289]
    reads = {0: po.Reads(*synth.short_reads_numpy(lengths[0], synth.n_reads_for(lengths[0]), 3)),
             2: H.random_reads(rng, lengths[2], 900, max_len=400),
             3: H.random_reads(rng, lengths[3], 6000, max_len=90, long_reads=True),
             4: H.random_reads(rng, lengths[4], 20000, max_len=60)}     # deep: several record batches per tile
    with DepthEngine(0) as eng:
        # stream 1: the streaming kernel over the records as they arrived (the default); 0: the tile kernel
        eng.set_option(OPT_FAST_KERNEL, 1 if stream else 0)
        eng.set_params(window_size=W, min_mapq=1, min_cov=4)
        eng.set_path(PATH_TILE)
        eng.set_outputs(sums_only=True)
        eng.set_contigs(lengths)
        for t, r in reads.items():
            eng.push(t, r.pos, r.flag, r.mapq, r.cigar_off, r.cigar)
        eng.compute()
        assert eng.stats().tile_kernel == {1: E.TK_SUMS_STREAM_RAW, 0: E.TK_TILE_SUMS}[stream]
        for t, L in enumerate(lengths):
            want = po.perbase_c(reads.get(t, H.empty_reads()), 1, 0, L)
            ws, _ = H.oracle_windows(want, W)
            assert np.array_equal(eng.window_sums(t), ws), (W, t)
        with pytest.raises(GdError):
            eng.windows(0)                                   # minima were not produced
        with pytest.raises(GdError):
            eng.callable_runs(0)
        with pytest.raises(GdError):
            eng.perbase(0)
        tids = np.array([[0]], np.int32)
        cells_s = eng.depthwed(tids, 1000)[0]
        eng.set_outputs(perbase=False)
        eng.compute()
        assert np.array_equal(eng.depthwed(tids, 1000)[0], cells_s)
        assert np.array_equal(eng.windows(0)[0], H.oracle_windows(po.perbase_c(reads[0], 1, 0, lengths[0]), W)[0])


def test_sums_only_small_windows_fall_back():
    from goleft_amd.engine import DepthEngine
    rng = np.random.default_rng(4)
    L = 20_000
    r = H.random_reads(rng, L, 3000, max_len=120)
    with DepthEngine(0) as eng:
        eng.set_params(window_size=13, min_mapq=0, min_cov=4)
        eng.set_outputs(sums_only=True)
        eng.set_contigs([L])
        eng.push(0, r.pos, r.flag, r.mapq, r.cigar_off, r.cigar)
        eng.compute()
        want = po.perbase_c(r, 0, 0, L)
        assert np.array_equal(eng.window_sums(0), H.oracle_windows(want, 13)[0])
        assert len(eng.callable_runs(0)) > 0                 # W < 32: the regular windows-only kernel ran


@pytest.mark.parametrize("seed", range(3))
def test_regions_batch_equals_single_calls_and_oracle(auto_eng, seed):
    """gd_regions (the --bed mode in one call): windows and class runs of many regions -- on several
    contigs, one of length 0 and one without reads, empty regions, regions past the contig end, one
    large region -- equal the single-region calls and the oracle's reductions of the per-base vector."""
    eng = auto_eng
    rng = np.random.default_rng(700 + seed)
    lens = [50_000, 0, 20_000, 9_000]
    contigs = [("r%d" % i, l) for i, l in enumerate(lens)]
    reads = {0: H.random_reads(rng, lens[0], 4000), 3: H.random_reads(rng, lens[3], 900)}
    W = int(rng.choice([1, 37, 250, 1000]))
    mincov, maxmean = int(rng.integers(1, 6)), int(rng.choice([0, 12]))
    run_engine(eng, contigs, reads, window_size=W, min_mapq=1, min_cov=mincov, max_mean_depth=maxmean)
    tids, starts, ends = [], [], []
    for _ in range(300):
        t = int(rng.integers(0, 4))
        a = int(rng.integers(0, max(1, lens[t]) + 50))
        b = a + int(rng.choice([0, 1, 7, 120, 600, 5000]))
        tids.append(t); starts.append(a); ends.append(b)
    tids.append(0); starts.append(3); ends.append(lens[0] + 200)           # one large region
    sums, mins, runs = eng.regions(tids, starts, ends)
    assert len(sums) == len(tids)
    per = {t: (po.perbase_c(reads.get(t, H.empty_reads()), 1, 0, l) if l else np.zeros(0, np.int32))
           for t, l in enumerate(lens)}
    for k, (t, a, b) in enumerate(zip(tids, starts, ends)):
        if b == a:
            assert len(sums[k]) == 0 and len(runs[k]) == 0
            continue
        d = np.zeros(b - a, np.int32)
        hi = min(b, lens[t])
        if hi > a:
            d[:hi - a] = per[t][a:hi]
        ws, wm = H.oracle_windows(d, W, a)
        assert np.array_equal(sums[k], ws), (k, t, a, b)
        assert np.array_equal(mins[k], wm if lens[t] else np.zeros(len(wm), np.int32)), (k, t, a, b)
        assert np.array_equal(runs[k], H.oracle_runs(d, mincov, maxmean, 1 << 62, a)), (k, t, a, b)
        if k % 17 == 0:                                                    # and the one-region API
            s1, m1 = eng.region_windows(t, a, b)
            assert np.array_equal(s1, sums[k]) and np.array_equal(m1, mins[k])
            assert np.array_equal(eng.region_callable(t, a, b), runs[k])


def test_slow_list_counter_survives_jobs_without_a_tile_table(auto_eng):
    """The straight-line tile kernel's slow list is counted in one of two alternating device counters, each
    zeroed by the tile-table kernel of the PREVIOUS compute; a compute in between that builds no tile table
    (scatter path, streaming sums) must not leave the counter -- and with it stale tiles of an earlier job --
    behind (found by tests/test_gpu_soak.py: a memory fault two jobs later)."""
    from goleft_amd.engine import PATH_AUTO, PATH_TILE, PATH_SCATTER
    eng = auto_eng
    rng = np.random.default_rng(77)
    try:
        for between in ("scatter", "sums"):
            a = {0: H.random_reads(rng, 12289, 3000), 1: H.random_reads(rng, 4097, 800)}
            eng.set_path(PATH_TILE)
            run_engine(eng, [("a", 12289), ("b", 4097)], a, window_size=1000, min_mapq=1, min_cov=4)
            assert eng.stats().n_slow_tiles >= 2
            lens = [4097, 4096, 64, 8192, 63]
            b = {t: H.random_reads(rng, L, 500) for t, L in enumerate(lens)}
            if between == "scatter":
                eng.set_path(PATH_SCATTER)
            else:
                eng.set_outputs(sums_only=True)
            run_engine(eng, [("c%d" % t, L) for t, L in enumerate(lens)], b, window_size=64, min_mapq=1, min_cov=4)
            eng.set_outputs(perbase=False)
            eng.set_path(PATH_AUTO)
            lens = [1, 12289, 12289, 0, 12289]
            c = {t: H.random_reads(rng, L, 600) for t, L in enumerate(lens) if L}
            contigs = [("d%d" % t, L) for t, L in enumerate(lens)]
            run_engine(eng, contigs, c, window_size=250, min_mapq=4, min_cov=4)
            assert eng.stats().n_slow_tiles == 4                    # the clipped last tile of each non-empty contig
            for t, L in enumerate(lens):
                want = po.perbase_c(c.get(t, H.empty_reads()), 4, 0, L) if L else np.zeros(0, np.int32)
                ws, wm = H.oracle_windows(want, 250)
                gs, gm = eng.windows(t)
                assert np.array_equal(gs, ws) and np.array_equal(gm, wm)
                assert np.array_equal(eng.callable_runs(t), H.oracle_runs(want, 4, 0, po.step_for(250)))
    finally:
        eng.set_outputs(perbase=True)
        eng.set_path(PATH_AUTO)


def test_adopted_device_records_are_checked():
    """gd_adopt_device gives device arrays the check gd_commit gives a host block: positions out of order are
    GD_E_UNSORTED, CSR offsets that decrease, do not start at 0 or end past the op array GD_E_INVALID -- before
    any kernel indexes with them; well-formed arrays (also with equal positions and reads without ops) pass."""
    import torch
    from goleft_amd.engine import DepthEngine, GdError
    from tests.test_gpu_soak import _adopt
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(41)
    L = 50_000
    r = H.random_reads(rng, L, 3000)

    def adopt(eng, pos, off, cigar=None):
        t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).view(dt)).to(dev)
        eng.adopt_device(0, t(pos.astype(np.int32), np.int32), t(r.flag, np.int16), t(r.mapq, np.uint8),
                         t(off.astype(np.uint32), np.int32), t(r.cigar if cigar is None else cigar, np.int32))

    with DepthEngine(0) as eng:
        eng.set_params(window_size=100, min_mapq=1, min_cov=4)
        eng.set_contigs([L])
        bad_pos = r.pos.copy(); bad_pos[1500], bad_pos[1501] = r.pos[-1], r.pos[0]
        with pytest.raises(GdError) as ei:
            adopt(eng, bad_pos, r.cigar_off)
        assert ei.value.status == -7
        for mutate in ("decreasing", "start", "end"):
            off = r.cigar_off.astype(np.int64).copy()
            if mutate == "decreasing":
                off[2000] = off[1999] - 1 if off[1999] > 0 else off[2001] + 5
                off[2000] = max(off[2000], off[2001] + 1)
            elif mutate == "start":
                off[0] = 1
            else:
                off[-1] = len(r.cigar) + 7
            with pytest.raises(GdError) as ei:
                adopt(eng, r.pos, off)
            assert ei.value.status == -1, mutate
        _adopt(eng, torch, 0, r)                                             # the well-formed arrays
        eng.compute()
        assert np.array_equal(eng.perbase(0), po.perbase_c(r, 1, 0, L))


def test_negative_positions_are_refused_at_the_boundary():
    """A placed BAM record has POS >= 0 (-1 means "no position"): gd_push / gd_commit and gd_adopt_device answer
    GD_E_RANGE instead of counting from a position that does not exist; the same records without the bad one pass."""
    import torch
    from goleft_amd.engine import DepthEngine, GdError
    L = 20000
    rng = np.random.default_rng(1)
    good = H.random_reads(rng, L, 500)
    bad = po.Reads(np.concatenate([[-3], good.pos]).astype(np.int32), np.concatenate([[0], good.flag]).astype(np.uint16),
                   np.concatenate([[60], good.mapq]).astype(np.uint8),
                   np.concatenate([[0], good.cigar_off + 1]).astype(np.uint32),
                   np.concatenate([[(10 << 4)], good.cigar]).astype(np.uint32))
    dev = torch.device("cuda", 0)
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).view(dt)).to(dev)

    def adopt(e, r):
        e.adopt_device(0, t(r.pos, np.int32), t(r.flag, np.int16), t(r.mapq, np.uint8), t(r.cigar_off, np.int32),
                       t(r.cigar, np.int32))

    with DepthEngine(0) as e:
        e.set_params(window_size=100, min_mapq=1, min_cov=1)
        e.set_contigs([L])
        with pytest.raises(GdError) as ei:
            e.push(0, bad.pos, bad.flag, bad.mapq, bad.cigar_off, bad.cigar)
        assert ei.value.status == -5
        with pytest.raises(GdError) as ei:
            adopt(e, bad)
        assert ei.value.status == -5
        want = po.perbase_c(good, 1, 0, L)
        e.set_contigs([L])
        e.push(0, good.pos, good.flag, good.mapq, good.cigar_off, good.cigar)
        e.compute()
        assert np.array_equal(e.perbase(0), want)
        e.set_contigs([L])
        adopt(e, good)
        e.compute()
        assert np.array_equal(e.perbase(0), want)
