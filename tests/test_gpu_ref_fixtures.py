"""-m gpu: the reference's OWN fixture files through the product, both decoders.

tests/golden/ref/ holds byte copies of /root/reference/depth/test/{t,hla,t-empty}.bam(.bai),
windows.bed, hg19.fa(.fai), fake.fa(.fai) -- htslib-written BGZF (libdeflate/zlib block
splitting, dynamic-Huffman headers as htslib emits them), a samtools-written .bai, real FASTA
line wrapping.  Every `goleft depth` invocation of depth/functional-test.sh:45-119 is run on those
exact files twice -- once reading the BAM on the device (gd_inflate_kernel + gd_bam_walk_kernel,
the default with a .bai) and once through the host decoder (GOLEFT_GPU_DECODE=0) -- and the BED
files are compared byte for byte with each other and with the CPU oracle's rows (+ the --stats
columns the oracle's restatement of faidx.Stats gives).  Plus: gd_inflate_bgzf on the raw files
against zlib, and gd_ingest_bgzf's record streams against the committed decoded streams.

The reference's own value check (depth/test/cmp.py: window mean within 0.5 of a live
`samtools depth -a -Q 1`) needs a samtools binary; tools/check_vs_samtools.sh is that check as
one command for the day one exists.
"""
import os
import zlib

import numpy as np
import pytest

from oracle import bamio, pyoracle as po
from tests import helpers as H

pytestmark = pytest.mark.gpu
REF = os.path.join(H.GOLDEN, "ref")


def fasta(path):
    """{name: bases} of a small FASTA (test infrastructure; the product reads it through the .fai)."""
    seqs, name = {}, None
    for line in open(path, "rb"):
        line = line.rstrip(b"\r\n")
        if line.startswith(b">"):
            name = line[1:].split()[0].decode()
            seqs[name] = []
        elif name is not None:
            seqs[name].append(line)
    return {k: b"".join(v) for k, v in seqs.items()}


def fai_contigs(path):
    return [(l.split("\t")[0], int(l.split("\t")[1])) for l in open(path) if l.strip()]


def oracle_beds(bam_key, fai, W, regions=None, stats_fa=None, **kw):
    """The oracle's two BED files for a run tiled by `fai` (depth.go:134-155) or by BED rows."""
    contigs, reads, _ = H.load_golden_bam(bam_key)
    names = [c[0] for c in contigs]
    if regions is None:
        fc = fai_contigs(fai)
        by = {i: reads[names.index(n)] for i, (n, _) in enumerate(fc) if n in names and names.index(n) in reads}
        hd, ca = po.depth_run_oracle(fc, by, W=W, **kw)
    else:
        hd, ca = po.depth_run_oracle(contigs, reads, W=W, regions=regions, **kw)
    if stats_fa:
        seqs = fasta(stats_fa)
        width = {l.split("\t")[0]: int(l.split("\t")[3]) for l in open(stats_fa + ".fai") if l.strip()}
        rows = []
        for line in hd.splitlines():
            chrom, s, e, _ = line.split("\t")
            # the default contract (GDH_STATS_FAIDX), line breaks as the FASTA has them
            rows.append(line + po.stats_columns(seqs[chrom], int(s), int(e), po.STATS_FAIDX, width[chrom]) + "\n")
        hd = "".join(rows)
    return hd, ca


def run_both(args, prefix):
    """Runs `goleft depth` with the device decoder and with the host decoder; returns the four texts
    after asserting that the two decoders wrote identical files."""
    from goleft_amd import depth
    out = {}
    old = os.environ.get("GOLEFT_GPU_DECODE")
    try:
        for mode in ("1", "0"):
            os.environ["GOLEFT_GPU_DECODE"] = mode
            p = "%s_dec%s" % (prefix, mode)
            rc = depth.Main([str(a) for a in args] + ["--prefix", p])
            assert rc == 0, (mode, args)
            out[mode] = (open(p + ".depth.bed").read(), open(p + ".callable.bed").read())
    finally:
        if old is None:
            os.environ.pop("GOLEFT_GPU_DECODE", None)
        else:
            os.environ["GOLEFT_GPU_DECODE"] = old
    assert out["1"] == out["0"], "device and host BAM decoders disagree"
    return out["1"]


def windows_bed_regions():
    return [po.chrom_start_end_c(l) for l in open(os.path.join(REF, "windows.bed"), "rb") if l.strip()]


# ---- depth/functional-test.sh:45-70: whole genome, --stats, hg19.fa, t.bam ---------------------
@pytest.mark.parametrize("W", [100, 1000000000, 55, 60, 71, 13, 2001])
def test_functional_wgs(tmp_path, W):
    fa = os.path.join(REF, "hg19.fa")
    got = run_both(["-Q", 1, "--ordered", "--windowsize", W, "--stats", "--reference", fa,
                    os.path.join(REF, "t.bam")], tmp_path / "x")
    want = oracle_beds("t", fa + ".fai", W, stats_fa=fa, Q=1, mincov=4)
    assert got[0] == want[0]
    assert got[1] == want[1]


# ---- :73-97: --bed test/windows.bed ------------------------------------------------------------
@pytest.mark.parametrize("W", [10, 1000000, 50, 55, 60, 71, 13, 2002])
def test_functional_bed(tmp_path, W):
    fa = os.path.join(REF, "hg19.fa")
    got = run_both(["--bed", os.path.join(REF, "windows.bed"), "-Q", 1, "--ordered", "--windowsize", W,
                    "--stats", "--reference", fa, os.path.join(REF, "t.bam")], tmp_path / "x")
    want = oracle_beds("t", None, W, regions=windows_bed_regions(), stats_fa=fa, Q=1, mincov=4)
    assert got[0] == want[0]
    assert got[1] == want[1]


# ---- :102-115: header-only BAM -----------------------------------------------------------------
def test_functional_empty(tmp_path):
    fa = os.path.join(REF, "hg19.fa")
    bam = os.path.join(REF, "t-empty.bam")
    got = run_both(["--windowsize", 10, "--q", 1, "--mincov", 4, "--reference", fa, "--processes", 1,
                    "--stats", bam], tmp_path / "x")
    want = oracle_beds("t_empty", fa + ".fai", 10, stats_fa=fa, Q=1, mincov=4)
    assert got == want
    assert set(l.split("\t")[3] for l in got[1].splitlines()) == {"NO_COVERAGE"}
    got = run_both(["--bed", os.path.join(REF, "windows.bed"), "--windowsize", 10, "--q", 1, "--mincov", 4,
                    "--reference", fa, "--processes", 1, "--stats", bam], tmp_path / "y")
    want = oracle_beds("t_empty", None, 10, regions=windows_bed_regions(), stats_fa=fa, Q=1, mincov=4)
    assert got == want


# ---- :118-119: contig names with ':' and '*', all defaults, fake.fa has ONE of the BAM's two refs
def test_functional_hla(tmp_path):
    fa = os.path.join(REF, "fake.fa")
    got = run_both(["-r", fa, os.path.join(REF, "hla.bam")], tmp_path / "xx")
    want = oracle_beds("hla", fa + ".fai", 250, Q=1, mincov=4)
    assert got == want
    assert got[0].startswith("HLA-A*01:01:01:01\t0\t250\t") and "chr22" not in got[0]


# ---- the device BGZF inflate on htslib-written members vs zlib ---------------------------------
@pytest.mark.parametrize("name", ["t", "hla", "t-empty"])
def test_inflate_reference_bam_equals_zlib(name):
    from goleft_amd.engine import DepthEngine
    data = open(os.path.join(REF, name + ".bam"), "rb").read()
    want = bamio.bgzf_decompress(data)               # zlib.decompress per member + CRC/ISIZE check
    with DepthEngine(0) as eng:
        got, status = eng.inflate_bgzf(data)
    assert (status == 0).all(), status[status != 0]
    assert len(got) == len(want) and zlib.crc32(got) == zlib.crc32(want)
    assert got == want


# ---- the device record walk from the samtools-written .bai vs the decoded streams --------------
@pytest.mark.parametrize("name,key", [("t", "t"), ("hla", "hla")])
def test_ingest_reference_bam(name, key):
    from goleft_amd.engine import DepthEngine
    contigs, reads, _ = H.load_golden_bam(key)
    path = os.path.join(REF, name + ".bam")
    data = open(path, "rb").read()
    lin = bamio.read_bai_linear(path + ".bai")
    with DepthEngine(0) as eng:
        eng.set_params(window_size=100, min_mapq=1, min_cov=4)
        eng.set_contigs([c[1] for c in contigs])
        # what htslib wrote into the index about each reference (pseudo-bin 37450, SAMv1 5.2): n_mapped + n_unmapped
        # records -- a reference-held count the device walk must deliver (tests/test_bai_pins.py: the whole index)
        from tests.test_bai_pins import parse_bai
        idx, _ = parse_bai(path + ".bai")
        for tid in range(len(contigs)):
            if len(lin[tid]):
                n = eng.ingest_bgzf(tid, data, 0, lin[tid])
                assert n == reads[tid].n
                assert idx[tid]["meta"] is not None and n == idx[tid]["meta"][2] + idx[tid]["meta"][3]
            else:
                assert idx[tid]["meta"] is None or idx[tid]["meta"][2] + idx[tid]["meta"][3] == 0
        eng.compute()
        for tid, (_, L) in enumerate(contigs):
            r = reads.get(tid, H.empty_reads())
            assert np.array_equal(eng.perbase(tid), po.perbase_c(r, 1, 0, L))


def test_known_answers_on_reference_bam(tmp_path):
    """SURVEY.md section 4 known answers, from the real file through the device decoder."""
    from goleft_amd import depth
    fa = os.path.join(REF, "hg19.fa")
    p = str(tmp_path / "k")
    assert depth.Main(["-Q", "1", "--ordered", "-w", "1000", "--prefix", p, "-r", fa,
                       os.path.join(REF, "t.bam")]) == 0
    rows = open(p + ".depth.bed").read().splitlines()
    assert rows[:6] == ["chrM\t0\t1000\t1001", "chrM\t1000\t2000\t1563", "chrM\t2000\t3000\t918.3",
                        "chrM\t3000\t4000\t1099", "chrM\t4000\t5000\t1117", "chrM\t5000\t6000\t45.8"]
    ca = open(p + ".callable.bed").read().splitlines()
    assert ca[:3] == ["chrM\t0\t1\tNO_COVERAGE", "chrM\t1\t5077\tCALLABLE", "chrM\t5077\t16571\tNO_COVERAGE"]
    assert len([r for r in ca if r.startswith("chr22")]) == 145
