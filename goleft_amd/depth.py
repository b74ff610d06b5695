"""`goleft depth` -- Python mirror of the reference's entry point
(/root/reference/depth/depth.go:162-175 `Main`, flags :27-41).

The host logic lives in the C++ twin (csrc/host/depth_host.cpp, the reference
being compiled Go); this module only forwards argv to it, so tests read like
the reference's functional tests (depth/functional-test.sh):

    from goleft_amd import depth
    rc = depth.Main(["--windowsize", "100", "--prefix", "x", "--reference", "hg19.fa", "t.bam"])
"""
from __future__ import annotations

import ctypes as C
import sys
from typing import Sequence

from . import _hostlib


def Main(argv: Sequence[str]) -> int:
    """Runs `goleft depth <argv>`; returns the exit code (depth.go:174)."""
    av = [b"goleft depth"] + [a.encode() for a in argv]
    arr = (C.c_char_p * len(av))(*av)
    return int(_hostlib.load().gdh_depth_main(len(av), arr))


if __name__ == "__main__":
    sys.exit(Main(sys.argv[1:]))
