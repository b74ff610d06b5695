"""The record function of tools/synth_bam.cpp, restated in numpy -- TEST INFRASTRUCTURE (like everything under oracle/).

`goleft_amd/synth-bam` writes the BAM files that bench.py's `bam_file_scope` and tools/scope3.py read through the CLI.
Its records are a pure function of (seed, contig index, read rank); this module computes the same (pos, flag, MAPQ, CIGAR)
without writing or reading any file, so that the BED files the CLI makes of such a BAM can be checked against the
ORACLE (oracle/depth_oracle.c: per-base depth + the restated callback of /root/reference/depth/depth.go:238-364) instead of
against the product's other decoder.  SEQ, QUAL, read names and aux tags do not reach `samtools depth -Q` without `-q`
(/root/reference/depth/depth.go:45) and are not restated.  tests/test_synthbam_twin.py holds the twin against the files the
tool writes (read back with oracle/bamio.py)."""
from __future__ import annotations

import hashlib
import os
import tempfile
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import pyoracle as po

RL = 150
_M64 = (1 << 64) - 1


def _mix(x: np.ndarray) -> np.ndarray:
    """splitmix64's finaliser, as tools/synth_bam.cpp:mix (uint64 arithmetic wraps)."""
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9e3779b97f4a7c15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xbf58476d1ce4e5b9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94d049bb133111eb)
        return x ^ (x >> np.uint64(31))


def n_reads(length: int, cov: float) -> int:
    return int(float(length) * cov / RL)


def records(ctg: int, length: int, cov: float = 30.0, seed: int = 20, lo: int = 0, hi: int | None = None) -> po.Reads:
    """Reads [lo, hi) of contig number `ctg` (0-based position in the file) of `synth-bam OUT chrS ...,length,... cov seed`."""
    n = n_reads(length, cov)
    hi = n if hi is None else min(hi, n)
    span = length - RL if length - RL > 0 else 1
    stride = span // n if n and span // n > 0 else 1
    i = np.arange(lo, hi, dtype=np.uint64)
    with np.errstate(over="ignore"):
        base = np.uint64(((seed + ctg * 7919) * 0x100000001b3) & _M64)
        h = _mix(base + i)
    # (i * span) / n: below 2^63 for every contig a BAM can hold (n < 2^31 * cov / 150, span < 2^31)
    pos = (i * np.uint64(span)) // np.uint64(max(n, 1)) + h % np.uint64(stride)
    pos = np.minimum(pos, np.uint64(span)).astype(np.int32)
    kind = ((h >> np.uint64(20)) % np.uint64(10000)).astype(np.int64)
    k40 = (h >> np.uint64(40))
    k50 = (h >> np.uint64(50))
    nc = np.where(kind < 9200, 1, np.where(kind < 9700, 2, 3)).astype(np.int64)
    off = np.zeros(hi - lo + 1, np.int64)
    np.cumsum(nc, out=off[1:])
    cig = np.zeros(int(off[-1]), np.uint32)
    o0 = off[:-1]
    m1 = kind < 9200
    cig[o0[m1]] = RL << 4
    m2 = (kind >= 9200) & (kind < 9700)
    k = (1 + (k40[m2] % np.uint64(30))).astype(np.uint32)
    cig[o0[m2]] = (k << np.uint32(4)) | np.uint32(4)
    cig[o0[m2] + 1] = (np.uint32(RL) - k) << np.uint32(4)
    m3 = (kind >= 9700) & (kind < 9900)
    a = (20 + (k40[m3] % np.uint64(100))).astype(np.uint32)
    d = (1 + (k50[m3] % np.uint64(10))).astype(np.uint32)
    cig[o0[m3]] = a << np.uint32(4)
    cig[o0[m3] + 1] = (d << np.uint32(4)) | np.uint32(2)
    cig[o0[m3] + 2] = (np.uint32(RL) - a) << np.uint32(4)
    m4 = kind >= 9900
    a = (20 + (k40[m4] % np.uint64(100))).astype(np.uint32)
    ins = (1 + (k50[m4] % np.uint64(10))).astype(np.uint32)
    cig[o0[m4]] = a << np.uint32(4)
    cig[o0[m4] + 1] = (ins << np.uint32(4)) | np.uint32(1)
    cig[o0[m4] + 2] = (np.uint32(RL) - a - ins) << np.uint32(4)
    fr = ((h >> np.uint64(8)) % np.uint64(1000)).astype(np.int64)
    flag = np.where((h & np.uint64(1)) != 0, 99, 147).astype(np.uint16)
    flag |= np.where(fr < 50, 0x400, np.where(fr < 51, 0x100, np.where(fr < 52, 0x200, np.where(fr < 57, 0x800, 0)))).astype(np.uint16)
    mapq = np.where(((h >> np.uint64(12)) % np.uint64(100)) == 0, 0, 60).astype(np.uint8)
    return po.Reads(pos, flag, mapq, off.astype(np.uint32), cig)


def contig_names(n: int, first: str = "chrS"):
    return [first if k == 0 else "%s_%d" % (first, k + 1) for k in range(n)]


def expected_beds(lengths, cov: float = 30.0, seed: int = 20, W: int = 1000, Q: int = 1, mincov: int = 4, maxmean: int = 0,
                  threads: int = 0, first: str = "chrS", chrom: str | None = None):
    """What `goleft depth -w W` must write for the file `synth-bam OUT <first> <lengths> <cov> <seed>`: the two BED texts'
    SHA-256 (depth.bed, callable.bed) and their row counts, from the oracle alone -- per 10 Mb tile (depth/depth.go:150-154)
    the per-base vector of the reads that can reach it, then the restated callback; tiles in genome order.
    chrom: only that contig (`--chrom`, depth/depth.go:122-131)."""
    po.lib()
    names = contig_names(len(lengths), first)
    threads = threads or min(32, os.cpu_count() or 1)
    hd, hc = hashlib.sha256(), hashlib.sha256()
    rows = [0, 0]
    with tempfile.TemporaryDirectory(prefix="gd_expected_") as td:
        for ctg, (name, L) in enumerate(zip(names, lengths)):
            if chrom is not None and name != chrom:
                continue
            r = records(ctg, L, cov, seed)
            tiles = list(po.tiles_c(L, W))

            def one(j):
                s, e = tiles[j]
                lo = int(np.searchsorted(r.pos, max(0, s - 4096), "left"))       # (no read of this model spans 4 kb)
                hi = int(np.searchsorted(r.pos, e, "left"))
                d = po.perbase_c(r.slice(lo, hi), Q, s, e, diff=True)
                if e > L:
                    d[max(0, L - s):] = 0
                a, b = os.path.join(td, "%d.d" % j), os.path.join(td, "%d.c" % j)
                po.callback_c(name, s, e, d, W, mincov, maxmean, a, b)
                return a, b

            with ThreadPoolExecutor(threads) as ex:
                for a, b in ex.map(one, range(len(tiles))):
                    for path, h, k in ((a, hd, 0), (b, hc, 1)):
                        with open(path, "rb") as fh:
                            blob = fh.read()
                        h.update(blob)
                        rows[k] += blob.count(b"\n")
                        os.unlink(path)
            del r
    return {"bed_sha256": [hd.hexdigest(), hc.hexdigest()], "rows": rows}
