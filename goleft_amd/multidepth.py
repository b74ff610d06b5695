"""`multidepth` entry point mirroring /root/reference/multidepth/multidepth.go:51 main()
(flag parsing, defaults and exit codes live in the C++ host twin)."""
from __future__ import annotations

import ctypes as C
import sys

from . import _hostlib


def Main(argv, out_path=None) -> int:
    """argv: the arguments after the program name, e.g.
    ["-c", "chr20", "--mincov", "7", "a.bam", "b.bam"].  Blocks go to out_path (stdout when None)."""
    lib = _hostlib.load()
    args = [b"multidepth"] + [str(a).encode() for a in argv]
    arr = (C.c_char_p * len(args))(*args)
    return int(lib.gdh_multidepth_run(len(args), arr, out_path.encode() if out_path else None))


if __name__ == "__main__":
    sys.exit(Main(sys.argv[1:]))
