"""-m gpu: `goleft depth` end to end through the C++ host and the HIP engine,
shaped like the reference's functional tests (depth/functional-test.sh): real
BAM files (written here from the committed record streams, since the
reference's fixtures do not exist on the GPU box), the reference's flags, and
byte-for-byte comparison of both BED outputs with the CPU oracle's."""
import os
import subprocess

import numpy as np
import pytest

from oracle import bamio, pyoracle as po
from tests import helpers as H

pytestmark = pytest.mark.gpu
ROOT = H.ROOT


@pytest.fixture(scope="module")
def workdir(tmp_path_factory):
    d = tmp_path_factory.mktemp("cli")
    for name in ("t", "hla", "t_empty"):
        contigs, reads, _ = H.load_golden_bam(name)
        bamio.write_bam(str(d / (name + ".bam")), contigs, reads, unplaced=3)
        (d / (name + ".fa.fai")).write_text("".join("%s\t%d\t6\t60\t61\n" % c for c in contigs))
    beds = H.golden_beds()
    (d / "windows.bed").write_text("".join("%s\t%d\t%d\n" % tuple(r) for r in beds["t"]["regions"]))
    return d


@pytest.fixture(scope="module")
def indexed(workdir):
    """The same BAMs with a .bai next to them: `goleft-depth` then reads them entirely on the
    device (gd_ingest_bgzf) instead of through the host decoder."""
    d = workdir / "idx"
    d.mkdir(exist_ok=True)
    for name in ("t", "hla", "t_empty"):
        contigs, reads, _ = H.load_golden_bam(name)
        bamio.write_bam(str(d / (name + ".bam")), contigs, reads, unplaced=3, index=True)
        (d / (name + ".fa.fai")).write_text("".join("%s\t%d\t6\t60\t61\n" % c for c in contigs))
    return d


def run_depth(args):
    from goleft_amd import depth
    return depth.Main([str(a) for a in args])


def read(prefix, kind, chrom=""):
    return open("%s%s.%s.bed" % (prefix, "." + chrom if chrom else "", kind)).read()


@pytest.mark.parametrize("W", [100, 1000, 55, 60, 71, 13, 2001, 250, 1000000000])
def test_wgs_matches_oracle(workdir, W):
    # functional-test.sh:45-70 (check_wgs, big window, odd windows)
    beds = H.golden_beds()["t"]["wg_w%d" % W]
    prefix = workdir / ("wg%d" % W)
    rc = run_depth(["-Q", 1, "--ordered", "--windowsize", W, "--prefix", prefix,
                    "--reference", workdir / "t.fa", workdir / "t.bam"])
    assert rc == 0
    assert read(prefix, "depth") == beds["depth"]
    assert read(prefix, "callable") == beds["callable"]


@pytest.mark.parametrize("W", [10, 50, 55, 60, 71, 13, 2002, 1000000])
def test_bed_mode_matches_oracle(workdir, W):
    # functional-test.sh:73-97 (--bed test/windows.bed)
    beds = H.golden_beds()["t"]["bed_w%d" % W]
    prefix = workdir / ("bed%d" % W)
    rc = run_depth(["--bed", workdir / "windows.bed", "-Q", 1, "--ordered", "--windowsize", W,
                    "--prefix", prefix, workdir / "t.bam"])
    assert rc == 0
    assert read(prefix, "depth") == beds["depth"]
    assert read(prefix, "callable") == beds["callable"]


def test_empty_bam(workdir):
    # functional-test.sh:102-115
    beds = H.golden_beds()["t-empty"]
    prefix = workdir / "empty"
    rc = run_depth(["--windowsize", 13, "--q", 1, "--mincov", 4, "--reference", workdir / "t_empty.fa",
                    "--processes", 1, "--prefix", prefix, workdir / "t_empty.bam"])
    assert rc == 0
    assert read(prefix, "depth") == beds["wg_w13"]["depth"]
    assert read(prefix, "callable") == beds["wg_w13"]["callable"]
    prefix = workdir / "empty_bed"
    rc = run_depth(["--bed", workdir / "windows.bed", "--windowsize", 10, "--prefix", prefix,
                    workdir / "t_empty.bam"])
    assert rc == 0
    assert read(prefix, "depth") == beds["bed_w10"]["depth"]
    assert read(prefix, "callable") == beds["bed_w10"]["callable"]


def test_hla_contig_names_and_defaults(workdir):
    # functional-test.sh:118-119: contig names with ':' and '*', default flags (W=250)
    beds = H.golden_beds()["hla"]["wg_w250"]
    prefix = workdir / "hla"
    assert run_depth(["-r", workdir / "hla.fa", "--prefix", prefix, workdir / "hla.bam"]) == 0
    assert read(prefix, "depth") == beds["depth"]
    assert read(prefix, "callable") == beds["callable"]


def test_flags_maxmeandepth_mincov_q(workdir):
    beds = H.golden_beds()["t"]
    prefix = workdir / "mm"
    assert run_depth(["-m", 1500, "-r", workdir / "t.fa", "--prefix", prefix, workdir / "t.bam"]) == 0
    assert read(prefix, "depth") == beds["wg_w250_maxmean1500"]["depth"]
    assert read(prefix, "callable") == beds["wg_w250_maxmean1500"]["callable"]
    assert "EXCESSIVE_COVERAGE" in read(prefix, "callable")
    prefix = workdir / "q0"
    assert run_depth(["-Q", 0, "--mincov", 10, "-r", workdir / "t.fa", "--prefix", prefix,
                      workdir / "t.bam"]) == 0
    assert read(prefix, "depth") == beds["wg_w250_Q0_mincov10"]["depth"]
    assert read(prefix, "callable") == beds["wg_w250_Q0_mincov10"]["callable"]


def test_chrom_filter_names_outputs(workdir):
    # depth.go:145 filter, :382-388 output naming <prefix>.<chrom>.depth.bed
    beds = H.golden_beds()["t"]["wg_w1000"]
    prefix = workdir / "c22"
    assert run_depth(["-c", "chr22", "-w", 1000, "-r", workdir / "t.fa", "--prefix", prefix,
                      workdir / "t.bam"]) == 0
    want = "".join(l + "\n" for l in beds["depth"].splitlines() if l.startswith("chr22\t"))
    assert read(prefix, "depth", "chr22") == want
    want = "".join(l + "\n" for l in beds["callable"].splitlines() if l.startswith("chr22\t"))
    assert read(prefix, "callable", "chr22") == want


def test_cli_binary_and_errors(workdir):
    exe = os.path.join(ROOT, "goleft_amd", "goleft-depth")
    prefix = workdir / "bin"
    p = subprocess.run([exe, "depth", "-w", "1000", "-r", str(workdir / "t.fa"), "--prefix",
                        str(prefix), str(workdir / "t.bam")], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    assert read(prefix, "depth") == H.golden_beds()["t"]["wg_w1000"]["depth"]
    # missing --prefix: usage error like go-arg's p.Fail (exit status 255)
    p = subprocess.run([exe, str(workdir / "t.bam")], capture_output=True, text=True)
    assert p.returncode == 255 and "prefix" in p.stderr
    # unreadable BAM: fatal like the reference's pcheck
    p = subprocess.run([exe, "--prefix", str(workdir / "zz"), "-r", str(workdir / "t.fa"),
                        str(workdir / "nope.bam")], capture_output=True, text=True)
    assert p.returncode == 1


def test_synthetic_multi_tile_contig(workdir):
    """A contig longer than one 10 Mb tile: callable runs split at the tile edge (quirk Q1)."""
    from goleft_amd import synth
    L = 12_000_000
    r = po.Reads(*synth.short_reads_numpy(L, synth.n_reads_for(L, 8.0), 3))
    contigs = [("chrBig", L), ("chrTiny", 77)]
    bamio.write_bam(str(workdir / "big.bam"), contigs, {0: r})
    (workdir / "big.fa.fai").write_text("chrBig\t%d\t8\t60\t61\nchrTiny\t77\t99\t60\t61\n" % L)
    prefix = workdir / "big"
    assert run_depth(["-w", 1000, "-r", workdir / "big.fa", "--prefix", prefix, workdir / "big.bam"]) == 0
    hd, ca = po.depth_run_oracle(contigs, {0: r}, W=1000, Q=1, mincov=4)
    assert read(prefix, "depth") == hd
    assert read(prefix, "callable") == ca
    assert any(l.split("\t")[2] == "10000000" for l in ca.splitlines())


@pytest.mark.parametrize("W", [100, 1000, 13])
def test_device_decoder_wgs_matches_oracle(workdir, indexed, W, capfd):
    beds = H.golden_beds()["t"]["wg_w%d" % W]
    prefix = indexed / ("wg%d" % W)
    os.environ["GOLEFT_DEPTH_TIMING"] = "1"
    try:
        rc = run_depth(["-Q", 1, "--ordered", "--windowsize", W, "--prefix", prefix,
                        "--reference", indexed / "t.fa", indexed / "t.bam"])
    finally:
        del os.environ["GOLEFT_DEPTH_TIMING"]
    assert rc == 0
    assert '"decoder": "device"' in capfd.readouterr().err       # the GPU path really ran
    assert read(prefix, "depth") == beds["depth"]
    assert read(prefix, "callable") == beds["callable"]


def test_device_decoder_bed_chrom_hla_empty(workdir, indexed):
    beds = H.golden_beds()
    prefix = indexed / "bed"
    assert run_depth(["--bed", workdir / "windows.bed", "-Q", 1, "--windowsize", 55, "--prefix", prefix,
                      indexed / "t.bam"]) == 0
    assert read(prefix, "depth") == beds["t"]["bed_w55"]["depth"]
    assert read(prefix, "callable") == beds["t"]["bed_w55"]["callable"]
    prefix = indexed / "c22"
    assert run_depth(["-c", "chr22", "-w", 1000, "-r", indexed / "t.fa", "--prefix", prefix, indexed / "t.bam"]) == 0
    want = "".join(l + "\n" for l in beds["t"]["wg_w1000"]["depth"].splitlines() if l.startswith("chr22\t"))
    assert read(prefix, "depth", "chr22") == want
    prefix = indexed / "hla"
    assert run_depth(["-r", indexed / "hla.fa", "--prefix", prefix, indexed / "hla.bam"]) == 0
    assert read(prefix, "depth") == beds["hla"]["wg_w250"]["depth"]
    assert read(prefix, "callable") == beds["hla"]["wg_w250"]["callable"]
    prefix = indexed / "empty"
    assert run_depth(["--windowsize", 13, "--reference", indexed / "t_empty.fa", "--prefix", prefix,
                      indexed / "t_empty.bam"]) == 0
    assert read(prefix, "depth") == beds["t-empty"]["wg_w13"]["depth"]


def test_host_decoder_still_selectable(workdir, indexed, capfd):
    beds = H.golden_beds()["t"]["wg_w1000"]
    prefix = indexed / "hostdec"
    os.environ["GOLEFT_DEPTH_TIMING"] = "1"
    os.environ["GOLEFT_GPU_DECODE"] = "0"
    try:
        assert run_depth(["-Q", 1, "-w", 1000, "--prefix", prefix, "-r", indexed / "t.fa", indexed / "t.bam"]) == 0
    finally:
        del os.environ["GOLEFT_DEPTH_TIMING"], os.environ["GOLEFT_GPU_DECODE"]
    assert '"decoder": "host"' in capfd.readouterr().err
    assert read(prefix, "depth") == beds["depth"]


def test_device_decoder_reads_a_reference_in_parts(workdir, capfd):
    """Passes cut INSIDE a chromosome at .bai anchors (GOLEFT_INGEST_PART_MB, default 1 GB; _KB here so that a small
    file is cut dozens of times): every part's records are appended to the contig's arrays (gd_ingest_decode_part),
    the member that holds a cut is inflated by both neighbours -- and the BED files are those of the uncut read and
    of the oracle's restatement of depth/depth.go:238-364."""
    from goleft_amd import synth
    La, Lb = 6_000_000, 2_500_000
    ra = po.Reads(*synth.short_reads_numpy(La, synth.n_reads_for(La, 9.0), 5))
    rb = po.Reads(*synth.short_reads_numpy(Lb, synth.n_reads_for(Lb, 5.0), 6))
    contigs = [("chrA", La), ("chrEmpty", 5000), ("chrB", Lb)]
    bam = workdir / "parts.bam"
    bamio.write_bam(str(bam), contigs, {0: ra, 2: rb}, index=True, unplaced=3)
    (workdir / "parts.fa.fai").write_text("".join("%s\t%d\t6\t60\t61\n" % c for c in contigs))
    hd, ca = po.depth_run_oracle(contigs, {0: ra, 2: rb}, W=250, Q=1, mincov=4)
    size = os.path.getsize(bam)
    outs = {}
    for kb in (0, 64, 300, max(64, size // 3000)):
        prefix = workdir / ("parts%d" % kb)
        os.environ["GOLEFT_INGEST_PART_KB"] = str(kb)
        os.environ["GOLEFT_DEPTH_TIMING"] = "1"
        try:
            assert run_depth(["-w", 250, "-r", workdir / "parts.fa", "--prefix", prefix, bam]) == 0
        finally:
            del os.environ["GOLEFT_INGEST_PART_KB"], os.environ["GOLEFT_DEPTH_TIMING"]
        assert '"decoder": "device"' in capfd.readouterr().err
        outs[kb] = (read(prefix, "depth"), read(prefix, "callable"))
        assert outs[kb][0] == hd and outs[kb][1] == ca, kb


def test_device_decoder_output_does_not_depend_on_how_the_file_is_fed(workdir, capfd):
    """The knobs of the device BAM read that round 6's last day turned -- workgroups of the copy kernel that pulls the staged
    pieces over the link (GOLEFT_INGEST_COPY_GRID = GD_OPT_INGEST_COPY_GRID: 16 by default, the 512 of rounds 4-6 slowed
    every other kernel 2.6 times), threads that list the BGZF members, pread workers -- change when bytes arrive, never what
    comes out: the BED files of every setting are the oracle's."""
    from goleft_amd import synth
    La, Lb = 5_000_000, 1_200_000
    ra = po.Reads(*synth.short_reads_numpy(La, synth.n_reads_for(La, 12.0), 15))
    rb = po.Reads(*synth.short_reads_numpy(Lb, synth.n_reads_for(Lb, 7.0), 16))
    contigs = [("chrA", La), ("chrB", Lb)]
    bam = workdir / "fed.bam"
    bamio.write_bam(str(bam), contigs, {0: ra, 1: rb}, index=True, unplaced=2)
    (workdir / "fed.fa.fai").write_text("".join("%s\t%d\t6\t60\t61\n" % c for c in contigs))
    hd, ca = po.depth_run_oracle(contigs, {0: ra, 1: rb}, W=500, Q=1, mincov=4)
    for i, env in enumerate(({}, {"GOLEFT_INGEST_COPY_GRID": "1"}, {"GOLEFT_INGEST_COPY_GRID": "700"},
                             {"GOLEFT_LIST_THREADS": "1", "GOLEFT_PUSH_THREADS": "2"},
                             {"GOLEFT_LIST_THREADS": "16", "GOLEFT_PUSH_THREADS": "16", "GOLEFT_INGEST_COPY_GRID": "64"})):
        prefix = workdir / ("fed%d" % i)
        os.environ.update(env, GOLEFT_DEPTH_TIMING="1")
        try:
            assert run_depth(["-w", 500, "-r", workdir / "fed.fa", "--prefix", prefix, bam]) == 0, env
        finally:
            for k in list(env) + ["GOLEFT_DEPTH_TIMING"]:
                del os.environ[k]
        assert '"decoder": "device"' in capfd.readouterr().err
        assert read(prefix, "depth") == hd and read(prefix, "callable") == ca, env
