"""-m gpu: THE PIN, switched on by the box.  The reference defines truth as a live `samtools depth -Q 1`
(/root/reference/depth/depth.go:45, depth/test/cmp.py:8-12); the image this repository is built and judged in has no
samtools, so per-base parity is "unpinned" (DESIGN.md section 5).  On a box that HAS one -- found on PATH or named by
$SAMTOOLS, and not this repository's own samtools-shaped shim -- this test runs tools/check_vs_samtools.py on the
reference's fixture BAMs per tile (cmp.py's own loop) and compares the HIP engine's gd_perbase AND the oracle with it,
bit for bit; with the three default-filter constants of SURVEY 8(c) exercised (-Q 0 / 1, a max-mean-depth).  Without one
it skips and says why -- the reason line is the evidence that the switch exists."""
import os
import shutil
import subprocess
import sys

import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu


def real_samtools():
    """a samtools that is not goleft_amd/shim/samtools (the zero-change drop-in of SURVEY 8(b) option A)"""
    st = shutil.which(os.environ.get("SAMTOOLS", "samtools"))
    if not st:
        return None
    if os.path.realpath(st).startswith(os.path.realpath(H.ROOT) + os.sep):
        return None
    try:
        out = subprocess.run([st, "--version"], capture_output=True, text=True, timeout=20)
    except (OSError, subprocess.SubprocessError):
        return None
    return st if out.returncode == 0 and "samtools" in out.stdout.lower() else None


SAMTOOLS = real_samtools()
needs_samtools = pytest.mark.skipif(SAMTOOLS is None, reason="no samtools on PATH (or $SAMTOOLS) in this image: per-base parity against the "
                                    "reference's own yardstick (depth/test/cmp.py:8) stays unpinned; the test runs by itself on a box that has one")


@needs_samtools
@pytest.mark.parametrize("name", ["t", "hla"])
@pytest.mark.parametrize("q,w,m", [(1, 250, 0), (0, 1000, 0), (1, 100, 100)])
def test_per_base_depth_equals_live_samtools(name, q, w, m):
    bam = os.path.join(H.ROOT, "tests", "golden", "ref", name + ".bam")
    p = subprocess.run([sys.executable, os.path.join(H.ROOT, "tools", "check_vs_samtools.py"), "--engine", "both", "--also-a",
                        "-Q", str(q), "-w", str(w), "-m", str(m), "--samtools", SAMTOOLS, bam],
                       capture_output=True, text=True, timeout=1200)
    assert p.returncode == 0, "PARITY AGAINST samtools FAILED\n" + p.stdout[-4000:] + p.stderr[-2000:]
    assert "0 differing" in p.stdout


def test_the_switch_is_wired():
    """(runs everywhere) the tool answers 2 -- "no samtools" -- when there is none, and the repository's own shim is not
    mistaken for one"""
    shim = os.path.join(H.ROOT, "goleft_amd", "shim")
    env = dict(os.environ, PATH=shim + os.pathsep + "/nonexistent")
    env.pop("SAMTOOLS", None)
    code = ("import os, sys; sys.path.insert(0, %r); import importlib.util as u; "
            "s = u.spec_from_file_location('pin', %r); m = u.module_from_spec(s); s.loader.exec_module(m); "
            "print(m.real_samtools())" % (H.ROOT, os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120)
    assert p.stdout.strip() == "None", (p.stdout, p.stderr[-500:])
    if SAMTOOLS is None:
        q = subprocess.run([sys.executable, os.path.join(H.ROOT, "tools", "check_vs_samtools.py"), os.path.join(H.ROOT, "tests", "golden", "ref", "t.bam")],
                           capture_output=True, text=True, timeout=120)
        assert q.returncode == 2 and "unpinned" in q.stderr
