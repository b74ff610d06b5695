"""-m gpu: several engine contexts behind ONE `goleft depth` process (GOLEFT_DEVICES) -- the
in-process counterpart of the reference's worker pool (depth/depth.go:392-394) with the merge still
done by the main thread in input order (:394-421).  The GPU box has one device, so the contexts are
"virtual shards" on device 0 (GOLEFT_DEVICES=0,0,...): same threads, same LPT assignment, same merge
as on N devices.  Bar: BED files byte-identical to the single-context run and to the oracle."""
import os

import numpy as np
import pytest

from oracle import bamio, pyoracle as po
from tests import helpers as H

pytestmark = pytest.mark.gpu
REF = os.path.join(H.GOLDEN, "ref")


def run_depth(args, devices=None, decode=None):
    from goleft_amd import depth
    env = {"GOLEFT_DEVICES": devices, "GOLEFT_GPU_DECODE": decode}
    old = {k: os.environ.get(k) for k in env}
    try:
        for k, v in env.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        return depth.Main([str(a) for a in args])
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def both(prefix):
    return open(str(prefix) + ".depth.bed").read(), open(str(prefix) + ".callable.bed").read()


@pytest.fixture(scope="module")
def genome(tmp_path_factory):
    """Nine contigs of very different sizes (one empty, one of 1 bp, one spanning two 10 Mb tiles is too
    slow for the oracle here: 2.5 Mb at step granularity instead), coordinate sorted, with a .bai."""
    d = tmp_path_factory.mktemp("multi")
    rng = np.random.default_rng(2024)
    lens = [2_500_000, 900_001, 40_000, 1, 333_333, 5_000, 1_200_000, 77, 640_000]
    contigs = [("c%d" % i, l) for i, l in enumerate(lens)]
    reads = {t: H.random_reads(rng, l, max(1, l // 40), max_len=160) for t, l in enumerate(lens) if t not in (3, 5)}
    bamio.write_bam(str(d / "g.bam"), contigs, reads, unplaced=5, index=True)
    (d / "g.fa.fai").write_text("".join("%s\t%d\t6\t60\t61\n" % c for c in contigs))
    return d, contigs, reads


@pytest.mark.parametrize("devices", ["0,0", "0,0,0", "0,0,0,0,0,0,0,0", "0," * 15 + "0"])
@pytest.mark.parametrize("decode", ["1", "0"])
def test_virtual_shards_whole_genome(genome, tmp_path, devices, decode):
    d, contigs, reads = genome
    args = ["-w", 500, "-Q", 1, "-r", d / "g.fa", d / "g.bam"]
    assert run_depth(args + ["--prefix", tmp_path / "one"], devices=None, decode=decode) == 0
    assert run_depth(args + ["--prefix", tmp_path / "many"], devices=devices, decode=decode) == 0
    assert both(tmp_path / "one") == both(tmp_path / "many")
    hd, ca = po.depth_run_oracle(contigs, reads, W=500, Q=1, mincov=4)
    assert both(tmp_path / "many") == (hd, ca)


def test_virtual_shards_bed_mode_and_stats(tmp_path):
    """--bed rows alternate between contigs that live on different contexts (region batches are per
    context; output order is the input order) and --stats runs on the first context."""
    fa = os.path.join(REF, "hg19.fa")
    bed = tmp_path / "rows.bed"
    bed.write_text("chr22\t14250\t15500\nchrM\t100\t1000\nchr22\t1575\t15800\nchrM\t2000\t5000\nchrM\t1\t3\nchr22:5-9\n")
    args = ["--bed", bed, "-Q", 1, "--windowsize", 55, "--stats", "--reference", fa, os.path.join(REF, "t.bam")]
    assert run_depth(args + ["--prefix", tmp_path / "one"]) == 0
    assert run_depth(args + ["--prefix", tmp_path / "two"], devices="0,0") == 0
    assert both(tmp_path / "one") == both(tmp_path / "two")
    assert open(str(tmp_path / "two") + ".depth.bed").read().count("\n") > 100


def test_reference_fixture_two_contexts(tmp_path):
    fa = os.path.join(REF, "hg19.fa")
    args = ["-Q", 1, "--ordered", "--windowsize", 100, "--stats", "--reference", fa, os.path.join(REF, "t.bam")]
    assert run_depth(args + ["--prefix", tmp_path / "one"]) == 0
    assert run_depth(args + ["--prefix", tmp_path / "two"], devices="0,0") == 0
    assert both(tmp_path / "one") == both(tmp_path / "two")


def test_bad_device_list_is_an_error(tmp_path):
    fa = os.path.join(REF, "hg19.fa")
    args = ["-r", fa, "--prefix", tmp_path / "x", os.path.join(REF, "t.bam")]
    assert run_depth(args, devices="0,zero") == 1
    assert run_depth(args, devices="4000") == 1          # no such device: refused, no CPU fallback


def test_unterminated_last_line_is_dropped_like_the_reference(tmp_path):
    """Quirk Q6 (depth/depth.go:107-110, :137-140): ReadBytes/ReadString return the unterminated last
    line together with io.EOF and the loop breaks before using it."""
    fa = os.path.join(REF, "hg19.fa")
    bam = os.path.join(REF, "t.bam")
    bed = tmp_path / "r.bed"
    bed.write_text("chrM\t100\t1000\nchrM\t2000\t2100")              # no newline after the second row
    assert run_depth(["--bed", bed, "-w", 100, "--prefix", tmp_path / "b", bam]) == 0
    rows = open(str(tmp_path / "b") + ".depth.bed").read().splitlines()
    assert rows and all(int(r.split("\t")[2]) <= 1000 for r in rows)
    (tmp_path / "f.fa.fai").write_text("chrM\t16571\t6\t60\t61\nchr22\t20001\t16861\t60\t61")   # chr22 unterminated
    assert run_depth(["-w", 1000, "-r", tmp_path / "f.fa", "--prefix", tmp_path / "w", bam]) == 0
    assert "chr22" not in open(str(tmp_path / "w") + ".depth.bed").read()
