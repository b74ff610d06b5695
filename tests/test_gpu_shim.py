"""-m gpu: the `samtools`-named shim (goleft_amd/shim/samtools, SURVEY.md section 8b option A): the exact command the
reference spawns per tile (depth/depth.go:45: `echo '<region>'; samtools depth -Q q -d D -r '<region>' '<bam>'`) on
the reference's own fixture BAM, its text parsed the way getPosDepth (:202-221) parses it and pushed through the
line-by-line restatement of the reference's callback (:238-364) -- BASELINE.json config 1, "plumbing" -- must give
the BED rows the engine's own CLI gives for the same file."""
import os
import subprocess

import numpy as np
import pytest

from oracle import pyoracle as po
from tests import helpers as H

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM_DIR = os.path.join(ROOT, "goleft_amd", "shim")
REF = os.path.join(ROOT, "tests", "golden", "ref")


def _samtools(*args):
    env = dict(os.environ, PATH=SHIM_DIR + os.pathsep + os.environ.get("PATH", ""))
    return subprocess.run(["samtools"] + list(args), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)


def test_shim_lines_equal_the_oracle_per_base_vector():
    contigs, reads, _ = H.load_golden_bam("t")
    bam = os.path.join(REF, "t.bam")
    for tid, (name, length) in enumerate(contigs):
        if tid not in reads:
            continue
        for s, e in ((0, min(length, 3000)), (1000, min(length, 16000)), (0, length)):
            p = _samtools("depth", "-Q", "1", "-d", "2510", "-r", "%s:%d-%d" % (name, s + 1, e), bam)
            assert p.returncode == 0, p.stderr.decode()
            want = po.perbase_c(reads[tid], 1, s, e)
            got = np.zeros(e - s, np.int32)
            for line in p.stdout.decode().splitlines():
                c, pos, d = line.split("\t")
                assert c == name and int(d) > 0               # positions of depth 0 are not printed
                got[int(pos) - 1 - s] = int(d)
            assert np.array_equal(got, want), (name, s, e)
    # -a prints the zeros too; anything that is not `samtools depth` is refused
    name, length = contigs[max(reads)]
    p = _samtools("depth", "-a", "-Q", "1", "-r", "%s:1-500" % name, bam)
    assert p.returncode == 0 and len(p.stdout.decode().splitlines()) == 500
    assert _samtools("view", bam).returncode == 1
    assert _samtools("depth", "-q", "20", bam).returncode == 1


def test_unmodified_reference_flow_through_the_shim(tmp_path):
    """What an unmodified goleft does, restated: tile the contigs (depth.go:132-154), per tile run the shim, parse the
    lines, run the callback -- and compare the concatenated BEDs with `goleft-depth` on the same file."""
    from goleft_amd import depth as gdepth
    contigs, reads, _ = H.load_golden_bam("t")
    bam = os.path.join(REF, "t.bam")
    W, Q, mincov = 1000, 1, 4
    hd_all, ca_all = [], []
    for name, length in contigs:
        for s, e in po.tiles_for(length, W):
            p = _samtools("depth", "-Q", str(Q), "-d", "2510", "-r", "%s:%d-%d" % (name, s + 1, e), bam)
            assert p.returncode == 0, p.stderr.decode()
            d = np.zeros(e - s, np.int32)
            for line in p.stdout.decode().splitlines():
                toks = line.split("\t", 1)[1].split("\t")       # getPosDepth: after the chrom, [0] = pos, [1] = depth
                d[int(toks[0]) - 1 - s] = int(toks[1])
            hd, ca = po.callback_py(name, s, e, d, W, mincov, 0)
            hd_all += hd
            ca_all += ca
    prefix = str(tmp_path / "out")
    assert gdepth.Main(["-w", str(W), "-Q", str(Q), "--mincov", str(mincov), "--prefix", prefix,
                        "-r", os.path.join(REF, "hg19.fa"), bam]) == 0
    assert open(prefix + ".depth.bed").read() == "".join(x + "\n" for x in hd_all)
    assert open(prefix + ".callable.bed").read() == "".join(x + "\n" for x in ca_all)
