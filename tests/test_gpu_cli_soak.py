"""-m gpu: randomized end-to-end soak of `goleft depth` (C++ host + HIP engine) against the CPU oracle.

Each case writes a random coordinate-sorted BAM (with or without a .bai, so both the device decoder and the
host decoder run), picks the reference's flags at random -- whole-genome tiling, --bed rows in both of the
reference's region spellings (depth/depth.go:73-100), -c -- and a random number of engine contexts
(GOLEFT_DEVICES, virtual shards on device 0), and compares both BED files byte for byte with
oracle.depth_run_oracle.  GOLEFT_SOAK_CLI=<n> sets the number of cases (default 6)."""
import os

import numpy as np
import pytest

from oracle import bamio, pyoracle as po
from tests import helpers as H
from tests.test_gpu_soak import _contig_reads
from tests.test_gpu_multidevice import run_depth, both

pytestmark = pytest.mark.gpu
CASES = int(os.environ.get("GOLEFT_SOAK_CLI", "6"))


@pytest.mark.parametrize("case", range(CASES))
def test_cli_soak(case, tmp_path):
    rng = np.random.default_rng(31000 + case)
    n_ctg = int(rng.integers(1, 7))
    lens = [int(rng.choice([1, 63, 4096, 4097, 12289, int(rng.integers(1, 200000)), int(rng.integers(1, 200000))]))
            for _ in range(n_ctg)]
    hla = rng.random() < 0.3        # names with ':' and '*' (not with --bed: the reference's region regex splits them)
    names = ["HLA-%d*0%d:01" % (i, i) if hla and rng.random() < 0.5 else "c%d" % i for i in range(n_ctg)]
    contigs = list(zip(names, lens))
    reads = {}
    for t, L in enumerate(lens):
        r = _contig_reads(rng, L)
        if r is not None:
            reads[t] = r
    index = bool(rng.random() < 0.7)
    bam = tmp_path / "x.bam"
    bamio.write_bam(str(bam), contigs, reads, unplaced=int(rng.integers(0, 4)), level=int(rng.choice([0, 1, 6])),
                    index=index)
    (tmp_path / "x.fa.fai").write_text("".join("%s\t%d\t6\t60\t61\n" % c for c in contigs))
    W = int(rng.choice([1, 7, 55, 100, 250, 1000, 4096, 1000000]))
    if max(lens) > 50000 and W < 7:
        W = 55                                               # keep the text outputs small
    Q = int(rng.choice([0, 1, 20]))
    mincov = int(rng.integers(1, 9))
    maxmean = int(rng.choice([0, 0, 12, 400]))
    args = ["-w", W, "-Q", Q, "--mincov", mincov, "--ordered"]
    if maxmean:
        args += ["-m", maxmean]
    mode = str(rng.choice(["wgs", "wgs", "bed", "chrom"]))
    if hla and mode == "bed":
        mode = "wgs"
    regions, chrom = None, ""
    if mode == "bed":
        rows, regions = [], []
        for _ in range(int(rng.integers(1, 40))):
            t = int(rng.integers(0, n_ctg))
            s = int(rng.integers(0, lens[t]))
            e = min(lens[t], s + int(rng.choice([1, 2, 50, 700, 5000, 100000])))
            if rng.random() < 0.3 and e > s:
                rows.append("%s:%d-%d\n" % (names[t], s + 1, e))         # 1-based inclusive (depth.go:73-86)
            else:
                rows.append("%s\t%d\t%d\n" % (names[t], s, e))
            regions.append((names[t], s, e))
        (tmp_path / "rows.bed").write_text("".join(rows))
        args += ["--bed", tmp_path / "rows.bed"]
    else:
        args += ["-r", tmp_path / "x.fa"]
        if mode == "chrom":
            chrom = names[int(rng.integers(0, n_ctg))]
            args += ["-c", chrom]
    devices = str(rng.choice(["", "0,0", "0,0,0"])) or None
    decode = str(rng.choice(["1", "1", "0"]))
    prefix = tmp_path / "out"
    tag = (case, mode, W, Q, mincov, maxmean, lens, index, devices, decode)
    part_kb = str(rng.choice(["", "", "64", "128"]))          # references read in parts cut at .bai anchors
    if part_kb:
        os.environ["GOLEFT_INGEST_PART_KB"] = part_kb
    try:
        assert run_depth(args + ["--prefix", prefix, bam], devices=devices, decode=decode) == 0, tag
    finally:
        os.environ.pop("GOLEFT_INGEST_PART_KB", None)
    if mode == "chrom":
        t = names.index(chrom)
        hd, ca = po.depth_run_oracle(contigs, reads, W=W, Q=Q, mincov=mincov, maxmean=maxmean)
        hd = "".join(l + "\n" for l in hd.splitlines() if l.split("\t")[0] == chrom)
        ca = "".join(l + "\n" for l in ca.splitlines() if l.split("\t")[0] == chrom)
        got = (open("%s.%s.depth.bed" % (prefix, chrom)).read(), open("%s.%s.callable.bed" % (prefix, chrom)).read())
    else:
        hd, ca = po.depth_run_oracle(contigs, reads, W=W, Q=Q, mincov=mincov, maxmean=maxmean, regions=regions)
        got = both(prefix)
    assert got[0] == hd, tag + ("depth",)
    assert got[1] == ca, tag + ("callable",)
