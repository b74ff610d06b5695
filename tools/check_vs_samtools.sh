#!/bin/sh
# One-command conformance check against a real samtools (see tools/check_vs_samtools.py).
# The reference's own check is depth/test/cmp.py:8-12 (window means within 0.5 of
# `samtools depth -a -Q 1`); this one is per base and exact.
#   tools/check_vs_samtools.sh [--engine oracle|gpu|both] [-Q q] [-w W] [-m M] file.bam
here=$(cd "$(dirname "$0")" && pwd)
if ! command -v "${SAMTOOLS:-samtools}" >/dev/null 2>&1; then
    echo "check_vs_samtools: no samtools on PATH (set SAMTOOLS=/path/to/samtools); parity stays unpinned" >&2
    exit 2
fi
[ $# -eq 0 ] && set -- "$here/../tests/golden/ref/t.bam"
exec python3 "$here/check_vs_samtools.py" "$@"
