"""CPU: oracle/synthbam.py (the numpy restatement of tools/synth_bam.cpp's record function) against the files the tool
really writes, read back with the pure-Python BAM reader -- so that bench.py's `bam_file_scope` can hold the CLI's BED
files against the ORACLE's rows for the same records (`oracle_identical`), not only against the product's other decoder."""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

from oracle import bamio, pyoracle as po, synthbam
from tests import helpers as H

EXE = os.path.join(H.ROOT, "goleft_amd", "synth-bam")
pytestmark = pytest.mark.skipif(not os.path.exists(EXE), reason="goleft_amd/synth-bam is not built")


@pytest.mark.parametrize("lengths,cov,seed,env", [
    ([300000, 170000, 151], 30.0, 20, {}),
    ([250000], 12.5, 7, {"SYNTH_BAM_AUX": "1", "SYNTH_BAM_LEVEL": "6"}),
    ([2000000, 40000], 30.0, 3, {"SYNTH_BAM_AUX": "1"}),
])
def test_twin_equals_the_records_of_the_file(tmp_path, lengths, cov, seed, env):
    bam = str(tmp_path / "s.bam")
    info = json.loads(subprocess.check_output([EXE, bam, "chrS", ",".join(map(str, lengths)), str(cov), str(seed), "3"],
                                              env=dict(os.environ, **env)).decode())
    contigs, reads = bamio.read_bam(bam)[1:3]
    assert [c[0] for c in contigs] == synthbam.contig_names(len(lengths)) and [c[1] for c in contigs] == lengths
    total = 0
    for ctg, L in enumerate(lengths):
        want = synthbam.records(ctg, L, cov, seed)
        got = reads[ctg]
        assert want.n == synthbam.n_reads(L, cov) == got.n
        for f in ("pos", "flag", "mapq", "cigar_off", "cigar"):
            assert np.array_equal(getattr(want, f), getattr(got, f)), (ctg, f)
        assert (np.diff(want.pos) >= 0).all()
        part = synthbam.records(ctg, L, cov, seed, lo=want.n // 3, hi=want.n // 2)      # a slice is the same reads
        assert np.array_equal(part.pos, want.pos[want.n // 3:want.n // 2])
        total += want.n
    assert total == info["reads"]
    if env.get("SYNTH_BAM_AUX"):
        assert "tags" in info["records"] and info["inflated_bytes"] > 330 * total


def test_expected_beds_are_the_oracle_run_on_the_files_records(tmp_path):
    lengths, W = [1234567, 4000, 99], 1000
    bam = str(tmp_path / "s.bam")
    subprocess.check_output([EXE, bam, "chrS", ",".join(map(str, lengths)), "30", "20", "2"])
    contigs, reads = bamio.read_bam(bam)[1:3]
    d, c = po.depth_run_oracle(contigs, reads, W=W, Q=1, mincov=4)
    exp = synthbam.expected_beds(lengths, 30.0, 20, W=W, threads=3)
    assert exp["bed_sha256"] == [hashlib.sha256(d.encode()).hexdigest(), hashlib.sha256(c.encode()).hexdigest()]
    assert exp["rows"] == [d.count("\n"), c.count("\n")]
    one = synthbam.expected_beds(lengths, 30.0, 20, W=16384, chrom="chrS_2")
    d2, _ = po.depth_run_oracle(contigs[1:2], {0: reads[1]}, W=16384, Q=1, mincov=4)
    assert one["bed_sha256"][0] == hashlib.sha256(d2.encode()).hexdigest()
