#!/usr/bin/env python3
"""The BED files `goleft depth` must write for bench.py's deterministic synthetic BAM files, from the ORACLE alone
(oracle/synthbam.py: the numpy twin of tools/synth_bam.cpp's record function -> oracle/depth_oracle.c per-base depth ->
the restated callback of depth/depth.go:238-364): SHA-256 of depth.bed and callable.bed, committed as
tests/golden/synth_bam_expected.json so that bench.py asserts them (`bam_file_scope.oracle_identical`).
    python tools/expected_bed_sha.py [genome] [chr20-21] [chr1-2] [paper]      (CPU only; the genome takes minutes)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from goleft_amd import synth          # (contig lengths only)
from oracle import synthbam

OUT = os.path.join(ROOT, "tests", "golden", "synth_bam_expected.json")
names = list(synth.HG19_NAMES)
FILES = {"genome": list(synth.HG19_LENGTHS), "chr1-2": list(synth.HG19_LENGTHS[:2]),
         "chr20-21": [synth.HG19_LENGTHS[names.index("chr20")], synth.HG19_LENGTHS[names.index("chr21")]]}


def key(which, W, chrom=None):
    return "%s:cov30:seed20:w%d%s" % (which, W, ":chrom=" + chrom if chrom else "")


def main():
    want = sys.argv[1:] or ["chr20-21"]
    have = json.load(open(OUT)) if os.path.exists(OUT) else {}
    for w in want:
        t0 = time.perf_counter()
        if w == "paper":                   # indexcov/paper/cmp.sh:6: `goleft depth --chrom <first contig> -w 16384` on the genome file
            k, r = key("genome", 16384, "chrS"), synthbam.expected_beds(FILES["genome"], W=16384, chrom="chrS")
        else:
            k, r = key(w, 1000), synthbam.expected_beds(FILES[w], W=1000)
        r["oracle_seconds"] = round(time.perf_counter() - t0, 1)
        r["what"] = "sha256 of depth.bed / callable.bed from oracle/synthbam.py + oracle/depth_oracle.c (tools/expected_bed_sha.py)"
        have[k] = r
        print(k, r, flush=True)
        with open(OUT, "w") as fh:
            json.dump(have, fh, indent=1, sort_keys=True)
            fh.write("\n")


if __name__ == "__main__":
    main()
