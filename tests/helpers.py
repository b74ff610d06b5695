"""Shared helpers for the tests (test infrastructure; may use oracle/)."""
import json
import os

import numpy as np

from oracle import pyoracle as po

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_golden_bam(name):
    """-> (contigs[(name,len)], {tid: Reads}, npz)"""
    z = np.load(os.path.join(GOLDEN, name + "_bam.npz"))
    contigs = list(zip([str(x) for x in z["contig_names"]], [int(x) for x in z["contig_lens"]]))
    reads = {}
    for tid in range(len(contigs)):
        if "pos_%d" % tid in z:
            reads[tid] = po.Reads(z["pos_%d" % tid], z["flag_%d" % tid], z["mapq_%d" % tid],
                                  z["cigar_off_%d" % tid], z["cigar_%d" % tid])
    return contigs, reads, z


def golden_beds():
    return json.load(open(os.path.join(GOLDEN, "fixture_beds.json")))


def empty_reads():
    return po.Reads(np.zeros(0, np.int32), np.zeros(0, np.uint16), np.zeros(0, np.uint8),
                    np.zeros(1, np.uint32), np.zeros(0, np.uint32))


def random_reads(rng, length, n, max_ops=6, max_len=300, long_reads=False):
    """Random record stream with every CIGAR op, including zero-length ops,
    leading/trailing clips, N skips and reads hanging over the contig end."""
    pos = np.sort(rng.integers(0, max(1, length), size=n)).astype(np.int32)
    nops = rng.integers(0, max_ops + 1, size=n)
    off = np.zeros(n + 1, np.uint32)
    off[1:] = np.cumsum(nops)
    m = int(off[-1])
    ops = rng.choice(9, size=m, p=[0.55, 0.08, 0.1, 0.04, 0.08, 0.02, 0.01, 0.06, 0.06])
    lens = rng.integers(0, max_len, size=m)
    if long_reads:
        lens = np.where(rng.random(m) < 0.02, lens * 50, lens)
    cigar = ((lens.astype(np.uint32) << 4) | ops.astype(np.uint32)).astype(np.uint32)
    flag = rng.choice([0, 16, 99, 147, 0x400, 0x100, 0x200, 0x4, 0x800, 0x410], size=n).astype(np.uint16)
    mapq = rng.choice([0, 1, 5, 60], size=n).astype(np.uint8)
    return po.Reads(pos, flag, mapq, off, cigar)


def long_cigar_reads(rng, length, n_ops_list, max_step=40, skip_every=0):
    """One read per entry of n_ops_list with exactly that many CIGAR ops: M runs
    interleaved with I / D (and S / = / X / P / zero-length ops), the shape of long-read
    alignments; every skip_every-th read also carries one long N skip.  Reads start
    at sorted random positions in the first half of the contig."""
    n = len(n_ops_list)
    pos = np.sort(rng.integers(0, max(1, length // 2), size=n)).astype(np.int32)
    off = np.zeros(n + 1, np.uint32)
    off[1:] = np.cumsum(n_ops_list)
    m = int(off[-1])
    ops = rng.choice(9, size=m, p=[0.5, 0.2, 0.2, 0.0, 0.02, 0.0, 0.02, 0.03, 0.03])
    lens = rng.integers(0, max_step, size=m)
    if skip_every:
        for i in range(0, n, skip_every):
            if n_ops_list[i] > 2:
                k = int(off[i]) + int(rng.integers(1, n_ops_list[i] - 1))
                ops[k], lens[k] = 3, int(rng.integers(5000, 40000))
    cigar = ((lens.astype(np.uint32) << 4) | ops.astype(np.uint32)).astype(np.uint32)
    flag = rng.choice([0, 16, 0x800, 0x400], size=n, p=[0.5, 0.3, 0.1, 0.1]).astype(np.uint16)
    mapq = rng.choice([0, 20, 60], size=n, p=[0.05, 0.25, 0.7]).astype(np.uint8)
    return po.Reads(pos, flag, mapq, off, cigar)


def oracle_windows(depth, W, start=0):
    """(sums, mins) of W-anchored windows clipped to [start, start+len(depth))."""
    end = start + len(depth)
    sums, mins = [], []
    for k in range(start // W, (end - 1) // W + 1 if end > start else 0):
        s, e = max(start, k * W), min(end, (k + 1) * W)
        seg = depth[s - start:e - start]
        sums.append(int(seg.astype(np.int64).sum()))
        mins.append(int(seg.min()))
    return np.asarray(sums, np.int64), np.asarray(mins, np.int32)


def oracle_runs(depth, mincov, maxmean, step, start=0):
    """[(start,end,cls)] with breaks at class changes and at multiples of step."""
    cls = np.where(depth == 0, 0, np.where(depth < mincov, 1,
                   np.where((maxmean > 0) & (depth >= maxmean), 3, 2)))
    n = len(depth)
    if n == 0:
        return np.zeros((0, 3), np.int32)
    p = np.arange(start, start + n)
    brk = np.ones(n, bool)
    brk[1:] = (cls[1:] != cls[:-1]) | (p[1:] % step == 0)
    s = p[brk]
    e = np.append(s[1:], start + n)
    return np.stack([s, e, cls[brk]], 1).astype(np.int32)


def ref_span(r):
    """Reference bases each record's CIGAR consumes (M, D, N, =, X), per read."""
    consumes = np.array([1, 0, 1, 1, 0, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 0], np.int64)
    c = r.cigar.astype(np.int64)
    cs = np.concatenate([[0], np.cumsum(consumes[c & 15] * (c >> 4))])
    off = r.cigar_off.astype(np.int64)
    return cs[off[1:]] - cs[off[:-1]]
