"""-m gpu: canonical CIGARs (csrc/gd_normalize.hpp, built when records arrive) against the op-by-op
Python restatement, and the two tile kernels (straight-line for ordinary tiles + generic for the rest,
vs generic for every tile, vs no normalisation at all) against each other and the oracle, bit for bit."""
import numpy as np
import pytest

from oracle import pyoracle as po
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _needs_canonical(request):
    """Canonical records are an optional part of the build (csrc/Makefile: make CANONICAL=1; include/goleft_depth.h
    GD_FEATURE_CANONICAL): the tests of them run where they exist; the tests below that do not need them say so."""
    from goleft_amd.engine import has_canonical
    if not has_canonical() and "no_canonical_needed" not in request.keywords:
        pytest.skip("canonical records are not part of this build")


def _engine(lengths, reads, fast=1, norm=1, fused=1, **params):
    from goleft_amd import engine as E
    eng = E.DepthEngine(0)
    eng.set_option(E.OPT_FUSED_NORMALIZE, fused)     # 1: the one-pass kernel (default); 0: count / scan / write / index launches
    eng.set_option(E.OPT_NORMALIZE, norm)
    eng.set_option(E.OPT_FAST_KERNEL, fast)
    eng.set_params(**params)
    eng.set_path(1)                                      # GD_PATH_TILE
    eng.set_contigs(lengths)
    for t, r in reads.items():
        eng.push(t, r.pos, r.flag, r.mapq, r.cigar_off, r.cigar)
    return eng


@pytest.mark.parametrize("fused", [1, 0])
@pytest.mark.parametrize("seed", range(3))
def test_canonical_cigars_equal_restatement(seed, fused):
    rng = np.random.default_rng(seed)
    L = 200_000
    r = H.random_reads(rng, L, 30_000, max_ops=9, max_len=120)
    with _engine([L], {0: r}, fused=fused, window_size=100) as eng:
        eng.compute()
        off, cig = eng.canonical_cigars(0, r.n)
    woff, wcig = po.canonical_cigars(r)
    assert np.array_equal(off, woff)
    assert np.array_equal(cig, wcig)
    assert set(np.unique(cig & 0xf)) <= {0, 3} and (cig >> 4).min() >= 1
    # a read is its M runs: never two neighbouring ops of one kind, never a trailing N
    last = cig[off[1:][off[1:] > off[:-1]] - 1]
    assert (last & 0xf == 0).all()


def test_canonical_overflow_and_degenerate_reads():
    """Merged lengths past the 28-bit field split off full-length ops; reads of only D/N/I/S ops have none."""
    M, D, N, I, S = 0, 2, 3, 1, 4
    big = 0x0fffffff
    cig = lambda *ops: [(ln << 4) | op for op, ln in ops]
    reads = [cig((M, big), (M, 5), (I, 3), (M, big)),            # M run of 2 * big + 5
             cig((M, 10), (D, big), (N, big), (D, 7), (M, 1)),   # N run of 2 * big + 7 between two Ms
             cig((D, 5), (N, 5)), cig((S, 5), (I, 2)), [],      # nothing counted
             cig((D, 3), (M, 4), (D, 9)),                        # leading N kept, trailing N dropped
             cig((M, 0), (M, 7), (D, 0), (M, 2))]               # zero-length ops vanish
    off = np.cumsum([0] + [len(x) for x in reads]).astype(np.uint32)
    flat = np.asarray([x for rd in reads for x in rd], np.uint32)
    n = len(reads)
    r = po.Reads(np.arange(n, dtype=np.int32) * 3, np.zeros(n, np.uint16), np.full(n, 60, np.uint8), off, flat)
    with _engine([1000], {0: r}, window_size=100) as eng:
        eng.set_path(0)
        got = eng.canonical_cigars(0, n) if False else None
    from goleft_amd import engine as E
    with E.DepthEngine(0) as eng:
        eng.set_option(E.OPT_NORMALIZE, 1)
        eng.set_params(window_size=100)
        eng.set_path(1)
        eng.set_contigs([1000])
        eng.push(0, r.pos, r.flag, r.mapq, r.cigar_off, r.cigar)
        try:
            eng.compute()                                        # spans beyond the tile path's limit: refused ...
        except E.GdError:
            pass
        goff, gcig = eng.canonical_cigars(0, n)                  # ... but the canonical form is there
    woff, wcig = po.canonical_cigars(r)
    assert np.array_equal(goff, woff) and np.array_equal(gcig, wcig)
    assert list(np.diff(woff)) == [3, 5, 0, 0, 0, 2, 1]
    assert wcig[-1] == (9 << 4)


@pytest.mark.parametrize("W,mincov,maxmean,step", [(100, 4, 0, 0), (1000, 4, 0, 0), (37, 2, 25, 3700), (5000, 1, 0, 0)])
def test_fast_kernel_equals_generic_and_oracle(W, mincov, maxmean, step):
    """Several contigs incl. clipped last tiles, an exactly-T contig, a deep pile-up (more than one batch of
    reads per tile -> slow list), a tile with more ops than the staging area, multi-op reads past the queue."""
    rng = np.random.default_rng(99)
    lengths = [4096 * 9 + 1234, 4096, 4096 * 3, 50_000, 1, 30_000]
    reads = {0: H.random_reads(rng, lengths[0], 9000, max_ops=5, max_len=150),
             1: H.random_reads(rng, lengths[1], 700, max_ops=2, max_len=200),
             2: H.random_reads(rng, 2000, 6000, max_ops=3, max_len=100),          # pile-up in the first tile
             3: H.random_reads(rng, lengths[3], 8000, max_ops=9, max_len=60),     # many multi-op reads, many ops
             5: H.random_reads(rng, lengths[5], 3000, max_ops=1, max_len=300)}
    res = {}
    # straight-line kernel on canonical records / on the records as they arrived (default, and with normalisation
    # switched off), generic kernel on canonical / original records
    for key, (fast, norm) in {"fast": (1, 1), "fast-raw": (1, 2), "fast-raw0": (1, 0), "generic": (0, 1), "raw": (0, 0)}.items():
        with _engine(lengths, reads, fast=fast, norm=norm, window_size=W, min_mapq=1, min_cov=mincov,
                     max_mean_depth=maxmean, step=step) as eng:
            eng.compute()
            from goleft_amd import engine as E
            assert eng.stats().tile_kernel == {"fast": E.TK_FAST, "fast-raw": E.TK_FAST_RAW, "fast-raw0": E.TK_FAST_RAW,
                                               "generic": E.TK_GENERIC, "raw": E.TK_GENERIC}[key]
            res[key] = [(eng.perbase(t), eng.windows(t), eng.callable_runs(t)) for t in range(len(lengths))]
    for t, L in enumerate(lengths):
        d = po.perbase_c(reads.get(t, H.empty_reads()), 1, 0, L)
        ws, wm = H.oracle_windows(d, W)
        for key in res:
            pb, (s, m), runs = res[key][t]
            assert np.array_equal(pb, d), (key, t)
            assert np.array_equal(s, ws) and np.array_equal(m, wm), (key, t)
            assert np.array_equal(runs, res["raw"][t][2]), (key, t)


def test_normalisation_is_ingest_time_not_compute_time():
    from goleft_amd import engine as E, synth
    L = 3_000_000
    r = po.Reads(*synth.short_reads_numpy(L, synth.n_reads_for(L), 5))
    with E.DepthEngine(0) as eng:
        eng.set_option(E.OPT_NORMALIZE, 1)
        eng.set_profiling(True)
        eng.set_params(window_size=1000)
        eng.set_contigs([L])
        eng.push(0, r.pos, r.flag, r.mapq, r.cigar_off, r.cigar)
        eng.compute()                                   # committed records: normalised by the first compute
        first = eng.kernel_ms(E.K_NORM)
        assert first > 0.0
        eng.compute()
        assert eng.kernel_ms(E.K_NORM) == first         # ... and not again
        off, cig = eng.canonical_cigars(0, r.n)
        single = float((np.diff(off) == 1).mean())
        raw_single = float((np.diff(r.cigar_off) == 1).mean())
        assert single > raw_single + 0.04               # soft clips and insertions became single-op reads
        assert np.array_equal(eng.perbase(0), po.perbase_c(r, 1, 0, L))


@pytest.mark.parametrize("fused", [1, 0])
@pytest.mark.parametrize("seed", range(3))
def test_wave_walked_canonical_cigars_equal_restatement(seed, fused):
    """Reads with more than 24 ops are normalised by a whole wave (runs spanning the 64-op groups, leading and
    trailing D/N runs, zero-length ops, a long N skip); shorter ones in the same 64-read unit by one lane."""
    rng = np.random.default_rng(100 + seed)
    L = 4_000_000
    n_ops = [int(x) for x in rng.choice([0, 1, 7, 24, 25, 63, 64, 65, 127, 128, 129, 500, 4097, 9000], size=150)]
    r = H.long_cigar_reads(rng, L, n_ops, max_step=25, skip_every=7)
    # a read of D/N ops only, one that starts and ends with deletions, a pure-M run across three groups
    cig = r.cigar.copy()
    o = r.cigar_off
    for i, n in enumerate(n_ops):
        if n >= 129 and i % 3 == 0:
            cig[o[i]:o[i] + 3] = [(5 << 4) | 2, (6 << 4) | 3, (0 << 4) | 2]
            cig[o[i + 1] - 2:o[i + 1]] = [(9 << 4) | 2, (4 << 4) | 3]
        if n == 500 and i % 2 == 0:
            cig[o[i]:o[i + 1]] = (rng.integers(1, 30, size=n).astype(np.uint32) << 4) | rng.choice([0, 7, 8, 1], size=n).astype(np.uint32)
        if n == 127:
            cig[o[i]:o[i + 1]] = (rng.integers(0, 30, size=n).astype(np.uint32) << 4) | rng.choice([2, 3, 1, 4], size=n).astype(np.uint32)
    r = po.Reads(r.pos, r.flag, r.mapq, r.cigar_off, cig)
    from goleft_amd import engine as E
    with E.DepthEngine(0) as eng:
        eng.set_option(E.OPT_FUSED_NORMALIZE, fused)
        eng.set_params(window_size=1000)
        eng.set_path(3)                                      # GD_PATH_CHUNK
        eng.set_contigs([L])
        eng.push(0, r.pos, r.flag, r.mapq, r.cigar_off, r.cigar)
        eng.compute()                                        # default: deletion lists straight from the records
        st0 = eng.stats()
        got0 = eng.perbase(0)
        assert st0.path == 3 and st0.n_canonical_ops == 0
        eng.normalize()                                      # ... and from the canonical CIGARs
        eng.compute()
        off, cig_got = eng.canonical_cigars(0, r.n)
        st = eng.stats()
        got = eng.perbase(0)
        # (straight from the records every D/N op is its own entry; the canonical route merges neighbours)
        assert np.array_equal(got0, got) and st0.n_deletions >= st.n_deletions == int((cig_got & 0xf == 3).sum())
    woff, wcig = po.canonical_cigars(r)
    assert np.array_equal(off, woff)
    assert np.array_equal(cig_got, wcig)
    assert st.path == 3 and st.n_canonical_ops == len(wcig)
    assert np.array_equal(got, po.perbase_c(r, 1, 0, L))


def test_default_computes_from_the_records_as_they_arrived():
    """GD_OPT_NORMALIZE = 2 (default): the short-read tile path builds nothing before its first compute (the raw
    straight-line kernel); gd_normalize builds the canonical records on request, after which the canonical
    kernel runs; gd_drop_derived goes back.  Same results every time."""
    from goleft_amd import engine as E, synth
    L = 2_000_000
    r = po.Reads(*synth.short_reads_numpy(L, synth.n_reads_for(L), 11))
    want = po.perbase_c(r, 1, 0, L)
    with E.DepthEngine(0) as eng:
        eng.set_profiling(True)
        eng.set_params(window_size=1000)
        eng.set_contigs([L])
        eng.push(0, r.pos, r.flag, r.mapq, r.cigar_off, r.cigar)
        eng.compute()
        st = eng.stats()
        assert eng.kernel_ms(E.K_NORM) == 0.0 and st.tile_kernel == E.TK_FAST_RAW and st.n_canonical_ops == 0
        assert st.n_slow_tiles <= 2                      # the clipped last tile (+ at most a dense one)
        assert np.array_equal(eng.perbase(0), want)
        with pytest.raises(E.GdError):
            eng.canonical_cigars(0, r.n)
        eng.normalize()
        assert eng.kernel_ms(E.K_NORM) > 0.0
        eng.compute()
        st = eng.stats()
        assert st.tile_kernel == E.TK_FAST and st.n_canonical_ops > 0
        assert np.array_equal(eng.perbase(0), want)
        off, cig = eng.canonical_cigars(0, r.n)
        woff, wcig = po.canonical_cigars(r)
        assert np.array_equal(off, woff) and np.array_equal(cig, wcig)
        eng.normalize(force=True)                        # again, into the same device block
        eng.compute()
        assert np.array_equal(eng.perbase(0), want)
        eng.drop_derived()
        eng.compute()
        assert eng.stats().tile_kernel == E.TK_FAST_RAW
        assert np.array_equal(eng.perbase(0), want)
        s, m = eng.windows(0)
        assert np.array_equal(s, want.reshape(-1, 1000).sum(1)) and np.array_equal(m, want.reshape(-1, 1000).min(1))


@pytest.mark.parametrize("fused", [1, 0])
def test_normalize_batches_many_contigs(fused):
    """One gd_normalize over contigs of every shape (empty, one read, short, long-read shaped, zero length): the
    batch's kernels find each 64-read unit's contig in the table; canonical CIGARs equal the restatement."""
    from goleft_amd import engine as E
    rng = np.random.default_rng(3)
    lengths = [70_000, 0, 5, 300_000, 64, 4096, 1_000_000, 129]
    reads = {0: H.random_reads(rng, lengths[0], 5000, max_ops=7),
             2: H.random_reads(rng, lengths[2], 1, max_ops=3),
             3: H.long_cigar_reads(rng, lengths[3], [int(x) for x in rng.choice([3, 30, 200, 1500], size=300)], skip_every=5),
             5: H.random_reads(rng, lengths[5], 64, max_ops=4),
             6: H.long_cigar_reads(rng, lengths[6], [int(x) for x in rng.choice([50, 700, 5000], size=130)]),
             7: H.random_reads(rng, lengths[7], 65, max_ops=2)}
    with E.DepthEngine(0) as eng:
        eng.set_option(E.OPT_FUSED_NORMALIZE, fused)
        eng.set_params(window_size=100)
        eng.set_contigs(lengths)
        for t, r in reads.items():
            eng.push(t, r.pos, r.flag, r.mapq, r.cigar_off, r.cigar)
        eng.normalize()
        for t, r in reads.items():
            off, cig = eng.canonical_cigars(t, r.n)
            woff, wcig = po.canonical_cigars(r)
            assert np.array_equal(off, woff) and np.array_equal(cig, wcig), t
        for path in (E.PATH_TILE, E.PATH_CHUNK, E.PATH_AUTO):
            eng.set_path(path)
            eng.compute()
            for t, L in enumerate(lengths):
                want = po.perbase_c(reads.get(t, H.empty_reads()), 1, 0, L) if L else np.zeros(0, np.int32)
                assert np.array_equal(eng.perbase(t), want), (path, t)


def test_position_index_of_the_fused_pass_serves_sparse_and_clustered_records():
    """The one-pass normalisation derives the position index from the positions it loads anyway: reads in clusters
    with megabase gaps between them, many reads at one position, a long empty tail, a contig without records -- the
    tile kernel on canonical records (its tile table is looked up in that index) must still be bit exact."""
    from goleft_amd import engine as E
    rng = np.random.default_rng(12)
    L = 9_000_000
    pos = np.sort(np.concatenate([np.full(700, 5), rng.integers(0, 3000, 900), rng.integers(2_500_000, 2_500_400, 5000),
                                  np.full(3, 4_194_304), rng.integers(6_000_000, 6_300_000, 2000)])).astype(np.int32)
    n = len(pos)
    r = po.Reads(pos, np.zeros(n, np.uint16), np.full(n, 60, np.uint8), np.arange(n + 1, dtype=np.uint32),
                 np.full(n, (100 << 4) | 0, np.uint32))
    lengths = [L, 70_000, L]
    for fused in (1, 0):
        with E.DepthEngine(0) as eng:
            eng.set_option(E.OPT_FUSED_NORMALIZE, fused)
            eng.set_params(window_size=1000)
            eng.set_path(E.PATH_TILE)
            eng.set_contigs(lengths)
            eng.push(0, r.pos, r.flag, r.mapq, r.cigar_off, r.cigar)
            eng.push(2, r.pos[:1], r.flag[:1], r.mapq[:1], r.cigar_off[:2], r.cigar[:1])
            eng.normalize()
            eng.compute()
            assert eng.stats().tile_kernel == E.TK_FAST
            assert np.array_equal(eng.perbase(0), po.perbase_c(r, 1, 0, L))
            assert int(eng.perbase(1).sum()) == 0
            assert int(eng.perbase(2).sum()) == 100


def test_long_read_lists_when_deletions_outnumber_their_slots():
    """The long-read path's deletion lists straight from the records keep every D/N op as its own entry at the dense
    offset (op offset >> 1) + r -- room for half the ops.  Reads whose D/N ops outnumber that (D N D N ..., runs of
    deletions, nothing but deletions, a deletion tail) are walked again with merged runs, by the whole wave (more than
    24 ops) or one lane: same depth as the oracle either way, next to ordinary reads whose slots they must not touch."""
    from goleft_amd import engine as E
    M, I, D, N, S = 0, 1, 2, 3, 4
    cg = lambda *ops: [(ln << 4) | op for op, ln in ops]
    reads = [cg((M, 50)) + cg((D, 3), (N, 2)) * 150 + cg((M, 40)),                 # 302 ops, 300 of them D/N
             cg((M, 10)) + cg((D, 1)) * 200,                                        # a tail of deletions only
             cg((D, 2)) * 90,                                                       # nothing counted at all
             cg((S, 5), (M, 30), (D, 4), (D, 0), (N, 6), (D, 1), (M, 3)),           # short: 4 D/N of 7 ops
             cg((M, 20), (I, 2), (M, 20)) * 40,                                     # ordinary long read, no deletions
             cg((N, 7), (D, 7)) * 30 + cg((M, 9)) + cg((D, 1), (M, 1)) * 60,        # leading run, then alternating
             cg((M, 100))]
    off = np.cumsum([0] + [len(x) for x in reads]).astype(np.uint32)
    flat = np.asarray([x for rd in reads for x in rd], np.uint32)
    n = len(reads)
    pos = np.asarray([100, 150, 200, 900, 1000, 1500, 5000], np.int32)
    r = po.Reads(pos, np.zeros(n, np.uint16), np.full(n, 60, np.uint8), off, flat)
    L = 20_000
    want = po.perbase_c(r, 1, 0, L)
    for norm in (2, 1):
        with E.DepthEngine(0) as eng:
            eng.set_option(E.OPT_NORMALIZE, norm)
            eng.set_params(window_size=100)
            eng.set_path(E.PATH_CHUNK)
            eng.set_contigs([L])
            eng.push(0, r.pos, r.flag, r.mapq, r.cigar_off, r.cigar)
            eng.compute()
            assert np.array_equal(eng.perbase(0), want), norm
