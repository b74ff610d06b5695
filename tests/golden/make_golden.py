#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz and *.json from the reference's fixture BAMs.

Runs anywhere (the fixture files are committed under tests/golden/ref/):
    python tests/golden/make_golden.py

What is stored
  * the decoded record streams (pos/flag/mapq/cigar SoA per contig) of
    depth/test/{t,hla,t-empty}.bam -- derived data, decoded with
    oracle/bamio.py (the files themselves: tests/golden/ref/);
  * the oracle's per-base depth and BED outputs for those streams.  These are
    REGRESSION vectors of the oracle, not reference-pinned truth (the reference
    has no golden outputs for this path);
  * the survey-derived known answers that the oracle was
    checked against when this file was generated.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import bamio, pyoracle as po  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(OUT, "ref")   # byte copies of /root/reference/depth/test/* (fixture DATA, not sources)


def pack(contigs, reads):
    d = {"contig_names": np.asarray([c[0] for c in contigs]),
         "contig_lens": np.asarray([c[1] for c in contigs], np.int64)}
    for tid, r in reads.items():
        d["pos_%d" % tid] = r.pos
        d["flag_%d" % tid] = r.flag
        d["mapq_%d" % tid] = r.mapq
        d["cigar_off_%d" % tid] = r.cigar_off
        d["cigar_%d" % tid] = r.cigar
    return d


def bed_regions(path):
    regs = []
    for line in open(path, "rb"):
        if not line.strip():
            continue
        regs.append(po.chrom_start_end_c(line))
    return regs


def main():
    answers = {}
    for name in ("t", "hla", "t-empty"):
        text, contigs, reads, total = bamio.read_bam(os.path.join(REF, name + ".bam"))
        d = pack(contigs, reads)
        d["n_records_total"] = np.asarray(total)
        for tid, (cn, cl) in enumerate(contigs):
            if tid in reads:
                pb = po.perbase_c(reads[tid], 1, 0, cl)
                assert (pb == po.perbase_numpy(reads[tid], 1, 0, cl)).all()
                d["perbase_Q1_%d" % tid] = pb
        np.savez_compressed(os.path.join(OUT, name.replace("-", "_") + "_bam.npz"), **d)
        beds = {}
        for W in (1000, 250, 100, 55, 60, 71, 13, 2001, 1000000000):
            hd, ca = po.depth_run_oracle(contigs, reads, W=W, Q=1, mincov=4)
            beds["wg_w%d" % W] = {"depth": hd, "callable": ca}
        if name != "hla":
            regs = bed_regions(os.path.join(REF, "windows.bed"))
            for W in (10, 50, 55, 60, 71, 13, 2002, 1000000):
                hd, ca = po.depth_run_oracle(contigs, reads, W=W, Q=1, mincov=4, regions=regs)
                beds["bed_w%d" % W] = {"depth": hd, "callable": ca}
            beds["regions"] = regs
        hd, ca = po.depth_run_oracle(contigs, reads, W=250, Q=1, mincov=4, maxmean=1500)
        beds["wg_w250_maxmean1500"] = {"depth": hd, "callable": ca}
        hd, ca = po.depth_run_oracle(contigs, reads, W=250, Q=0, mincov=10)
        beds["wg_w250_Q0_mincov10"] = {"depth": hd, "callable": ca}
        answers[name] = beds

    # SURVEY.md section 4 known answers (throw-away decoder, not reference-pinned)
    t = np.load(os.path.join(OUT, "t_bam.npz"))
    m, c22 = t["perbase_Q1_0"], t["perbase_Q1_1"]
    assert int(m.sum()) == 5743876 and int(m.max()) == 2012 and int(m.argmax()) == 1289
    assert int((m > 0).sum()) == 5076
    assert int(c22.sum()) == 23813 and int(c22.max()) == 39 and int(c22.argmax()) == 15325
    assert int((c22 > 0).sum()) == 9811
    rows = answers["t"]["wg_w1000"]["depth"].splitlines()
    assert rows[:6] == ["chrM\t0\t1000\t1001", "chrM\t1000\t2000\t1563", "chrM\t2000\t3000\t918.3",
                        "chrM\t3000\t4000\t1099", "chrM\t4000\t5000\t1117", "chrM\t5000\t6000\t45.8"]
    crow = [r for r in rows if r.startswith("chr22")]
    assert crow[:4] == ["chr22\t0\t1000\t0.2", "chr22\t1000\t2000\t0.743",
                        "chr22\t2000\t3000\t0.697", "chr22\t3000\t4000\t1.271"]
    ca = answers["t"]["wg_w1000"]["callable"].splitlines()
    assert ca[:3] == ["chrM\t0\t1\tNO_COVERAGE", "chrM\t1\t5077\tCALLABLE",
                      "chrM\t5077\t16571\tNO_COVERAGE"]
    assert len([r for r in ca if r.startswith("chr22")]) == 145
    json.dump(answers, open(os.path.join(OUT, "fixture_beds.json"), "w"), indent=0)
    print("golden written to", OUT)


if __name__ == "__main__":
    main()
