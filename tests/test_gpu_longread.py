"""-m gpu: the long-read path's structures (csrc/gd_chunk.hpp) at their edges -- deletion lists whose D/N ops outnumber
their slots (the merging fallback walks), and the tile index that the same pass fills (round 5): boundaries that a
deletion starts exactly on, reads that begin or end on one, ops that cross many of them at once, and reads whose index
slots are too few (three ops spanning 100 kb: PT_SEARCH, the tile kernel bisects the list) -- next to ordinary long
reads whose slots they must not touch.  Every case against the oracle's per-base vector, bit for bit."""
import numpy as np
import pytest

from oracle import pyoracle as po

pytestmark = pytest.mark.gpu

M, I, D, N, S, H, P, EQ, X = 0, 1, 2, 3, 4, 5, 6, 7, 8


def cg(*ops):
    return [(ln << 4) | op for op, ln in ops]


def reads_of(cigars, positions, flags=None, mapqs=None):
    off = np.cumsum([0] + [len(x) for x in cigars]).astype(np.uint32)
    flat = np.asarray([x for rd in cigars for x in rd], np.uint32)
    n = len(cigars)
    order = np.argsort(np.asarray(positions), kind="stable")
    assert (order == np.arange(n)).all(), "give the reads in coordinate order"
    return po.Reads(np.asarray(positions, np.int32), np.zeros(n, np.uint16) if flags is None else np.asarray(flags, np.uint16),
                    np.full(n, 60, np.uint8) if mapqs is None else np.asarray(mapqs, np.uint8), off, flat)


def check(r, L, W=100, **params):
    from goleft_amd import engine as E
    want = po.perbase_c(r, params.get("min_mapq", 1), 0, L)
    with E.DepthEngine(0) as eng:
        eng.set_params(window_size=W, **params)
        eng.set_path(E.PATH_CHUNK)
        eng.set_contigs([L])
        eng.push(0, r.pos, r.flag, r.mapq, r.cigar_off, r.cigar)
        eng.compute()
        got = eng.perbase(0)
        assert np.array_equal(got, want), (int((got != want).sum()), int(np.flatnonzero(got != want)[0]))
        sums, mins = eng.windows(0)
        nw = (L + W - 1) // W
        pad = np.zeros(nw * W, np.int64)
        pad[:L] = want
        assert np.array_equal(sums, pad.reshape(nw, W).sum(1))
        eng.rebuild_derived()                              # the same structures again, into the block they occupy
        eng.compute()
        assert np.array_equal(eng.perbase(0), want)
    return want


def test_deletions_that_outnumber_their_slots():
    """The deletion lists keep every D/N op as its own entry at the dense offset (op offset >> 1) + r -- room for half the
    ops.  Reads whose D/N ops outnumber that (D N D N ..., runs of deletions, nothing but deletions, a deletion tail) are
    walked again with merged runs, by the whole wave (more than 24 ops) or one lane; their tile index does not describe
    the merged list, so the tile kernel bisects it."""
    reads = [cg((M, 50)) + cg((D, 3), (N, 2)) * 150 + cg((M, 40)),                 # 302 ops, 300 of them D/N
             cg((M, 10)) + cg((D, 1)) * 200,                                        # a tail of deletions only
             cg((D, 2)) * 90,                                                       # nothing counted at all
             cg((S, 5), (M, 30), (D, 4), (D, 0), (N, 6), (D, 1), (M, 3)),           # short: 4 D/N of 7 ops
             cg((M, 20), (I, 2), (M, 20)) * 40,                                     # ordinary long read, no deletions
             cg((N, 7), (D, 7)) * 30 + cg((M, 9)) + cg((D, 1), (M, 1)) * 60,        # leading run, then alternating
             cg((M, 100))]
    check(reads_of(reads, [100, 150, 200, 900, 1000, 1500, 5000]), 20_000)


def test_tile_index_at_its_boundaries():
    """Entry k of a read's index = its deletions that start BEFORE boundary ((pos >> 12) + k) << 12.  Deletions that start
    exactly on a boundary, one base before and one after; a read that starts on a boundary, ends on one, ends one base
    past one; a deletion that spans a whole tile and more; walked by one lane (few ops) and by the wave (many)."""
    T = 4096
    few = [cg((M, T - 100), (D, 5), (M, 95), (D, 7), (M, 50)),                      # pos 100: deletions end / start at 4096
           cg((M, T - 1), (D, 3), (M, 10)),                                         # pos 4096: deletion starts at 8191
           cg((M, T), (D, 3), (M, 10)),                                             # pos 4097: deletion starts at 8193
           cg((M, 96), (D, 2), (M, 4000 - 2)),                                      # pos 8096: deletion exactly at 8192, read ends at 12192
           cg((M, 10), (D, 3 * T + 17), (M, 10)),                                   # a deletion over three tiles
           cg((M, 5), (D, 1), (M, 4096 - 6))]                                       # ends exactly on a boundary
    pos_few = [100, 4096, 4097, 8096, 9000, 3 * T + 8192]
    many_a = cg((S, 30)) + cg((M, 31), (D, 1), (M, 31), (I, 2)) * 400 + cg((M, 7))  # 1600 ops, a deletion every 63 bases: some on boundaries
    many_b = cg((M, 63), (D, 1)) * 300 + cg((M, 1))                                 # deletions at pos + 63 + 64 k: exactly on every boundary it meets
    cig = few + [many_a, many_b]
    pos = pos_few + [5 * T + 1, 6 * T + 1]
    order = np.argsort(pos, kind="stable")
    check(reads_of([cig[i] for i in order], [pos[i] for i in order]), 40 * T + 77, W=1000)


def test_reads_whose_index_slots_are_too_few():
    """A read owns (ops >> 6) + 3 index slots: enough whenever its ops average <= 64 reference bases.  Spliced reads -- three
    ops, 100 kb -- and a long read with one huge skip need more: no index (PT_SEARCH), the tile kernel bisects the list.
    Between ordinary long reads whose slots lie right behind theirs."""
    rng = np.random.default_rng(5)
    cig, pos = [], []
    p = 1000
    for k in range(60):
        kind = k % 4
        if kind == 0:                                                               # spliced: M N M, the skip over many tiles
            cig.append(cg((M, 50), (N, int(rng.integers(5000, 120000))), (M, 50)))
        elif kind == 1:                                                             # an ordinary long read
            cig.append(cg((M, 20), (D, 1), (M, 20), (I, 1)) * int(rng.integers(30, 200)) + cg((M, 5)))
        elif kind == 2:                                                             # many ops AND a skip that outgrows them
            cig.append(cg((M, 10), (D, 2)) * 40 + cg((N, 300000)) + cg((M, 10), (D, 2)) * 40 + cg((M, 3)))
        else:                                                                       # two skips and deletions between them: a list to bisect
            cig.append(cg((M, 30), (N, 20000)) + cg((M, 9), (D, 1)) * 20 + cg((N, 33000), (M, 30)))
        pos.append(p)
        p += int(rng.integers(0, 3000))
    check(reads_of(cig, pos), 600_000, W=250)


@pytest.mark.parametrize("seed", range(4))
def test_random_long_reads(seed):
    """Random long reads (20 .. 3000 ops: M / = / X runs, insertions, deletions, skips, clips), random filters."""
    rng = np.random.default_rng(100 + seed)
    L = 300_000
    n = 400
    pos = np.sort(rng.integers(0, L - 100, n)).astype(np.int32)
    cig = []
    for _ in range(n):
        k = int(rng.integers(20, 3000)) if rng.integers(0, 4) else int(rng.integers(1, 20))
        ops = rng.choice([M, M, M, EQ, X, I, D, D, N, S, P], k)
        lens = np.where(np.isin(ops, [N]), rng.integers(1, 3000, k), rng.integers(0, 40, k))
        cig.append([(int(l) << 4) | int(o) for o, l in zip(ops, lens)])
    flags = np.where(rng.integers(0, 10, n) == 0, 0x400, 0).astype(np.uint16)
    mapqs = np.where(rng.integers(0, 12, n) == 0, 0, 60).astype(np.uint8)
    check(reads_of(cig, pos, flags, mapqs), L, W=int(rng.choice([64, 250, 1000])))
