#!/usr/bin/env python3
"""Conformance hook: per-base depth of this repository vs a REAL `samtools depth`.

The arithmetic `goleft depth` relies on lives in an external samtools
(/root/reference/depth/depth.go:45); the reference's only value check is
depth/test/cmp.py:8-12 against a live binary.  No samtools exists in the build image or on the
GPU box, so per-base parity is UNPINNED.  This script is the pin, as one
command, for the day a samtools is on PATH:

    tools/check_vs_samtools.sh tests/golden/ref/t.bam            # oracle vs samtools (CPU only)
    tools/check_vs_samtools.sh --engine gpu  some.bam            # HIP engine vs samtools
    tools/check_vs_samtools.sh --engine both -Q 0 -w 1000 some.bam

For every W-aligned <=10 Mb tile goleft would generate (depth.go:122-159) it runs exactly the
child the reference runs (`samtools depth -Q q -d maxmean+2500 -r chr:s-e bam`), expands the text
to a per-base vector (omitted positions = 0) and compares it bit for bit with
  * oracle:  oracle/depth_oracle.c gdo_perbase on the decoded record stream, and/or
  * gpu:     gd_perbase of the HIP engine (records through the C++ host BAM reader + gd_push).
Exit status 0 = identical everywhere; 1 = a difference (first ten printed per tile); 2 = no
samtools.  `--also-a` repeats with `-a` (the form cmp.py uses) and checks that it only adds zeros.
"""
import argparse
import os
import shutil
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def samtools_perbase(samtools, bam, chrom, s, e, q, maxmean, extra=()):
    cmd = [samtools, "depth", *extra, "-Q", str(q), "-d", str(maxmean + 2500),
           "-r", "%s:%d-%d" % (chrom, s + 1, e), bam]
    out = subprocess.run(cmd, check=True, capture_output=True).stdout
    v = np.zeros(e - s, np.int32)
    if out:
        a = np.loadtxt(out.decode().splitlines(), dtype=np.int64, usecols=(1, 2), ndmin=2)
        v[a[:, 0] - 1 - s] = a[:, 1]
    return v


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("bam")
    ap.add_argument("--engine", choices=["oracle", "gpu", "both"], default="oracle")
    ap.add_argument("-Q", "--q", type=int, default=1)
    ap.add_argument("-w", "--windowsize", type=int, default=250)
    ap.add_argument("-m", "--maxmeandepth", type=int, default=0)
    ap.add_argument("--samtools", default=os.environ.get("SAMTOOLS", "samtools"))
    ap.add_argument("--also-a", action="store_true")
    a = ap.parse_args()
    st = shutil.which(a.samtools)
    if st and os.path.realpath(st).startswith(os.path.realpath(ROOT) + os.sep):
        st = None                                          # this repository's own samtools-shaped shim is not the yardstick
    if not st:
        print("check_vs_samtools: no `%s` on PATH -- per-base parity stays unpinned" % a.samtools, file=sys.stderr)
        return 2
    print(subprocess.run([st, "--version"], capture_output=True, text=True).stdout.splitlines()[0])

    from oracle import bamio, pyoracle as po
    _, contigs, reads, _ = bamio.read_bam(a.bam)
    eng = None
    if a.engine in ("gpu", "both"):
        from goleft_amd.engine import DepthEngine
        eng = DepthEngine(0)
        eng.set_params(window_size=a.windowsize, min_mapq=a.q, min_cov=4, max_mean_depth=a.maxmeandepth)
        eng.set_contigs([c[1] for c in contigs])
        for tid, r in reads.items():
            eng.push(tid, r.pos, r.flag, r.mapq, r.cigar_off, r.cigar)
        eng.compute()
    bad = tiles = 0
    for tid, (chrom, length) in enumerate(contigs):
        r = reads.get(tid)
        for s, e in po.tiles_c(length, a.windowsize):
            want = samtools_perbase(st, a.bam, chrom, s, e, a.q, a.maxmeandepth)
            if a.also_a:
                wa = samtools_perbase(st, a.bam, chrom, s, e, a.q, a.maxmeandepth, extra=("-a",))
                assert np.array_equal(want, wa), "-a changed non-zero depths in %s:%d-%d" % (chrom, s + 1, e)
            got = {}
            if a.engine in ("oracle", "both"):
                got["oracle"] = po.perbase_c(r, a.q, s, e) if r is not None else np.zeros(e - s, np.int32)
            if eng is not None:
                got["gpu"] = eng.perbase(tid, s, e)
            tiles += 1
            for name, g in got.items():
                d = np.flatnonzero(g != want)
                if len(d):
                    bad += 1
                    print("DIFF %s %s:%d-%d: %d positions" % (name, chrom, s + 1, e, len(d)))
                    for i in d[:10]:
                        print("   pos1 %d  samtools %d  %s %d" % (s + i + 1, want[i], name, g[i]))
    if eng is not None:
        eng.close()
    print("%d tiles checked, %d differing" % (tiles, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
