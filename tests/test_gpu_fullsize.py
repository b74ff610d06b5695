"""-m gpu: BASELINE.json config 2 at its FULL size (synthetic 30x chr20, 63 025 520 bp,
12.6 M reads) and the long-read path on a full chr20 (20x ONT-like, ~10^8 CIGAR ops), on the
device paths that serve them.  Checked two ways: directly against the C oracle (which still
finishes a chromosome in about a second), and through size-independent properties that need
no oracle at all -- conservation of counted bases, window sums / minima recomputed from the
per-base vector, class runs tiling the contig with breaks exactly at class changes and at
multiples of the step."""
import numpy as np
import pytest

from goleft_amd import synth
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu

L = synth.CHR20_LEN
W, Q, MINCOV = 1000, 1, 4


def counted_bases(r, q, length, flag_mask=0x704):
    """sum over kept reads of the M/=/X bases that fall inside [0, length): what the per-base
    vector must add up to (`samtools depth` counts nothing else, depth/depth.go:45)."""
    n = r.n
    keep = ((r.flag & flag_mask) == 0) & (r.mapq >= q)
    op = (r.cigar & 0xF).astype(np.int64)
    ln = (r.cigar >> 4).astype(np.int64)
    consumes = np.isin(op, (0, 2, 3, 7, 8))
    counted = np.isin(op, (0, 7, 8))
    read_of = np.repeat(np.arange(n), np.diff(r.cigar_off.astype(np.int64)))
    cons = np.where(consumes, ln, 0)
    before = np.cumsum(cons) - cons                           # consumed before this op, global
    first = r.cigar_off[:-1].astype(np.int64)
    base = np.where(np.diff(r.cigar_off.astype(np.int64)) > 0, before[np.minimum(first, len(before) - 1)], 0)
    start = r.pos.astype(np.int64)[read_of] + before - base[read_of]
    end = np.minimum(start + ln, length)
    return int(np.where(counted & keep[read_of], np.maximum(end - np.maximum(start, 0), 0), 0).sum())


def check_properties(eng, r, q, perbase):
    assert perbase.shape == (L,) and perbase.min() >= 0
    assert int(perbase.sum(dtype=np.int64)) == counted_bases(r, q, L)
    sums, mins = eng.windows(0)
    edges = np.arange(0, L, W)
    assert np.array_equal(sums, np.add.reduceat(perbase.astype(np.int64), edges))
    assert np.array_equal(mins, np.minimum.reduceat(perbase, edges))
    runs = eng.callable_runs(0)
    step = po.step_for(W)
    assert runs[0, 0] == 0 and runs[-1, 1] == L and np.array_equal(runs[1:, 0], runs[:-1, 1])
    cls = np.where(perbase == 0, 0, np.where(perbase < MINCOV, 1, 2)).astype(np.int8)
    brk = np.flatnonzero((cls[1:] != cls[:-1]) | (np.arange(1, L) % step == 0)) + 1
    assert np.array_equal(runs[1:, 0], brk)                   # breaks exactly there, nowhere else
    assert np.array_equal(runs[:, 2], cls[runs[:, 0]])


def test_config2_chr20_full_size():
    from goleft_amd.engine import DepthEngine, PATH_TILE
    n = synth.n_reads_for(L)
    assert n == 12605104
    r = po.Reads(*synth.short_reads_numpy(L, n, 20))
    with DepthEngine(0) as eng:
        eng.set_params(window_size=W, min_mapq=Q, min_cov=MINCOV)
        eng.set_path(PATH_TILE)
        eng.set_contigs([L])
        eng.push(0, r.pos, r.flag, r.mapq, r.cigar_off, r.cigar)
        eng.compute()
        got = eng.perbase(0)
        check_properties(eng, r, Q, got)
    assert np.array_equal(got, po.perbase_c(r, Q, 0, L, diff=True))


def test_ont_chr20_full_size_chunk_path():
    from goleft_amd.engine import DepthEngine, PATH_CHUNK
    n = synth.n_ont_reads_for(L, 20.0)
    r = po.Reads(*synth.ont_reads_numpy(L, n, 20))
    assert r.cigar.shape[0] > 50_000_000
    with DepthEngine(0) as eng:
        eng.set_params(window_size=W, min_mapq=Q, min_cov=MINCOV)
        eng.set_path(PATH_CHUNK)
        eng.set_contigs([L])
        eng.push(0, r.pos, r.flag, r.mapq, r.cigar_off, r.cigar)
        eng.compute()
        got = eng.perbase(0)
        check_properties(eng, r, Q, got)
    assert np.array_equal(got, po.perbase_c(r, Q, 0, L, diff=True))
