"""tools/check_vs_samtools.py is the one-command parity pin for the day a real samtools exists
(the reference's only value check is depth/test/cmp.py:8-12 against a live binary).  Here the hook
itself is tested: with a stand-in `samtools` that prints the text contract of depth.go:45 from the
oracle it must pass, with one that is off by one at a single position it must fail, and without a
samtools it must say so (exit 2).  This tests the HARNESS, not parity."""
import os
import stat
import subprocess
import sys

from tests import helpers as H

TOOL = os.path.join(H.ROOT, "tools", "check_vs_samtools.py")
BAM = os.path.join(H.GOLDEN, "ref", "t.bam")

SHIM = r'''#!%(py)s
import sys
sys.path.insert(0, %(root)r)
if sys.argv[1] == "--version":
    print("samtools 0.0-standin"); sys.exit(0)
assert sys.argv[1] == "depth"
a = sys.argv[2:]
all_pos = "-a" in a
q = int(a[a.index("-Q") + 1]); reg = a[a.index("-r") + 1]; bam = a[-1]
from oracle import bamio, pyoracle as po
chrom, s, e = po.chrom_start_end_c(reg.encode())
_, contigs, reads, _ = bamio.read_bam(bam)
tid = [c[0] for c in contigs].index(chrom)
d = po.perbase_c(reads[tid], q, s, e)
out = []
for i, v in enumerate(d):
    v = int(v) + (%(skew)d if s + i == 1289 else 0)
    if v or all_pos:
        out.append("%%s\t%%d\t%%d\n" %% (chrom, s + i + 1, v))
sys.stdout.write("".join(out))
'''


def shim(tmp_path, skew):
    p = tmp_path / "samtools"
    p.write_text(SHIM % {"py": sys.executable, "root": H.ROOT, "skew": skew})
    p.chmod(p.stat().st_mode | stat.S_IEXEC)
    return str(p)


def run(samtools, *extra):
    return subprocess.run([sys.executable, TOOL, "--samtools", samtools, "-w", "1000", *extra, BAM],
                          capture_output=True, text=True)


def test_hook_passes_on_agreeing_samtools(tmp_path):
    p = run(shim(tmp_path, 0), "--also-a")
    assert p.returncode == 0, p.stdout + p.stderr
    assert "0 differing" in p.stdout


def test_hook_flags_a_single_base_difference(tmp_path):
    p = run(shim(tmp_path, 1))
    assert p.returncode == 1
    assert "pos1 1290" in p.stdout


def test_hook_without_samtools(tmp_path):
    p = run(str(tmp_path / "no-such-samtools"))
    assert p.returncode == 2 and "unpinned" in p.stderr
