"""CPU: the C oracle against the golden vectors, an independent brute-force
counter, and a second restatement of the reference's callback."""
import os

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from oracle import pyoracle as po
from tests import helpers as H

REF_TEST = "/root/reference/depth/test"


def test_survey_known_answers_t_bam():
    # SURVEY.md section 4 (throw-away decoder numbers; not reference-pinned)
    contigs, reads, z = H.load_golden_bam("t")
    assert contigs == [("chrM", 16571), ("chr22", 20001)]
    assert reads[0].n == 80002 and reads[1].n == 264
    assert int(z["n_records_total"]) == 80330
    kept = sum(int((((r.flag & 0x704) == 0) & (r.mapq >= 1)).sum()) for r in reads.values())
    by_flag = sum(int(((r.flag & 0x704) != 0).sum()) for r in reads.values())
    # 76 054 kept; 4 208 placed + 64 unplaced = 4 272 dropped by flag, 4 by MAPQ
    assert (kept, by_flag + 64, 80266 - kept - by_flag) == (76054, 4272, 4)
    m = po.perbase_c(reads[0], 1, 0, 16571)
    c22 = po.perbase_c(reads[1], 1, 0, 20001)
    assert (int(m.sum()), int(m.max()), int(m.argmax()), int((m > 0).sum())) == (5743876, 2012, 1289, 5076)
    assert (int(c22.sum()), int(c22.max()), int(c22.argmax()), int((c22 > 0).sum())) == (23813, 39, 15325, 9811)
    assert np.array_equal(m, z["perbase_Q1_0"]) and np.array_equal(c22, z["perbase_Q1_1"])


@pytest.mark.parametrize("name", ["t", "hla", "t_empty"])
def test_c_oracle_matches_independent_counters(name):
    contigs, reads, _ = H.load_golden_bam(name)
    for tid, r in reads.items():
        clen = contigs[tid][1]
        full = po.perbase_c(r, 1, 0, clen)
        assert np.array_equal(full, po.perbase_numpy(r, 1, 0, clen))
        assert np.array_equal(full, po.perbase_c(r, 1, 0, clen, diff=True))
        # brute force (pure Python loops) on a slice of the reads and a sub-region
        sub = r.slice(0, min(r.n, 1500))
        s, e = 10, min(clen, 700)
        assert np.array_equal(po.perbase_bruteforce(sub, 1, s, e), po.perbase_c(sub, 1, s, e))
        assert np.array_equal(po.perbase_bruteforce(sub, 0, s, e, flag_mask=0),
                              po.perbase_c(sub, 0, s, e, flag_mask=0))


def test_golden_beds_regenerate():
    """The committed BED fixtures are what the oracle produces today."""
    beds = H.golden_beds()
    for name, key in (("t", "t"), ("hla", "hla"), ("t_empty", "t-empty")):
        contigs, reads, _ = H.load_golden_bam(name)
        for W in (1000, 13):
            hd, ca = po.depth_run_oracle(contigs, reads, W=W, Q=1, mincov=4)
            assert hd == beds[key]["wg_w%d" % W]["depth"]
            assert ca == beds[key]["wg_w%d" % W]["callable"]
    contigs, reads, _ = H.load_golden_bam("t")
    regs = [tuple(r) for r in beds["t"]["regions"]]
    hd, ca = po.depth_run_oracle(contigs, reads, W=55, Q=1, mincov=4, regions=regs)
    assert hd == beds["t"]["bed_w55"]["depth"] and ca == beds["t"]["bed_w55"]["callable"]


def test_quirk_q2_overlapping_window():
    # region chrM:5011-6000, W=200 on the fixture
    contigs, reads, _ = H.load_golden_bam("t")
    hd, _ = po.depth_run_oracle(contigs, reads, W=200, Q=1, mincov=4, regions=[("chrM", 5010, 6000)])
    rows = hd.splitlines()
    assert rows[0].startswith("chrM\t5010\t5210\t171.8")
    assert rows[1] == "chrM\t5200\t5400\t0"


def test_tiling_invariants_like_functional_test():
    # depth/functional-test.sh:45-70: outputs tile the .fai exactly, no duplicates
    beds = H.golden_beds()
    contigs, _, _ = H.load_golden_bam("t")
    for key in ("wg_w100", "wg_w55", "wg_w60", "wg_w71", "wg_w13", "wg_w2001", "wg_w1000000000"):
        for kind in ("depth", "callable"):
            rows = [r.split("\t") for r in beds["t"][key][kind].splitlines()]
            assert len(set(map(tuple, rows))) == len(rows)
            for name, clen in contigs:
                iv = [(int(r[1]), int(r[2])) for r in rows if r[0] == name]
                assert iv[0][0] == 0 and iv[-1][1] == clen
                assert all(a[1] == b[0] for a, b in zip(iv, iv[1:]))


def test_window_means_within_reference_tolerance():
    # depth/test/cmp.py:12: |mean(samtools depth -a) - printed| <= 0.5 per row
    beds = H.golden_beds()
    contigs, reads, z = H.load_golden_bam("t")
    names = [c[0] for c in contigs]
    for key in ("wg_w100", "bed_w50", "bed_w1000000"):
        for row in beds["t"][key]["depth"].splitlines():
            chrom, s, e, mean = row.split("\t")[:4]
            pb = z["perbase_Q1_%d" % names.index(chrom)]
            s, e = int(s), int(e)
            assert abs(float(pb[s:e].sum()) / (e - s) - float(mean)) <= 0.5


@pytest.mark.parametrize("line,want", [
    (b"chr22\t14250\t15500\n", ("chr22", 14250, 15500)),
    (b"chrM:1-16571\n", ("chrM", 0, 16571)),
    (b"HLA-A*01:01:01:01:1-16571\n", ("HLA-A*01:01:01:01", 0, 16571)),
    (b"chr1:0-5", ("chr1", 0, 5)),
    (b"a\t3\t9\tname\t0\t+\n", ("a", 3, 9)),
])
def test_region_parse(line, want):
    assert po.chrom_start_end_c(line) == want


def test_region_parse_failure():
    with pytest.raises(ValueError):
        po.chrom_start_end_c(b"no region here\n")


@pytest.mark.parametrize("W", [1, 13, 250, 1000, 9999999, 10000000, 10000001, 1000000000])
def test_tiles(W):
    assert po.lib().gdo_step(W) == po.step_for(W)
    for length in (1, 16571, 20001, 63025520):
        t = po.tiles_c(length, W)
        assert t == po.tiles_for(length, W)
        assert t[0][0] == 0 and t[-1][1] == length
        assert all(s % W == 0 for s, _ in t)


@settings(max_examples=60, deadline=None)
@given(st.integers(0, 2 ** 31), st.integers(1, 400), st.integers(1, 60), st.integers(0, 300),
       st.integers(1, 8), st.sampled_from([0, 5, 30]))
def test_callback_c_equals_callback_py(seed, length, W, start, mincov, maxmean):
    rng = np.random.default_rng(seed)
    depth = rng.integers(0, 12, size=length).astype(np.int32)
    depth[rng.random(length) < 0.4] = 0
    if rng.random() < 0.3:
        depth[int(rng.integers(0, length)):] = 0
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        hd, ca = os.path.join(td, "d"), os.path.join(td, "c")
        po.callback_c("chrZ", start, start + length, depth, W, mincov, maxmean, hd, ca)
        got_hd, got_ca = open(hd).read().splitlines(), open(ca).read().splitlines()
    want_hd, want_ca = po.callback_py("chrZ", start, start + length, depth, W, mincov, maxmean)
    assert got_hd == want_hd
    assert got_ca == want_ca


@settings(max_examples=40, deadline=None)
@given(st.integers(0, 2 ** 31))
def test_perbase_random_cigars(seed):
    rng = np.random.default_rng(seed)
    length = int(rng.integers(1, 3000))
    r = H.random_reads(rng, length, int(rng.integers(0, 120)))
    s = int(rng.integers(0, length))
    e = int(rng.integers(s, length + 1))
    want = po.perbase_bruteforce(r, 1, s, e)
    assert np.array_equal(po.perbase_c(r, 1, s, e), want)
    assert np.array_equal(po.perbase_c(r, 1, s, e, diff=True), want)
    assert np.array_equal(po.perbase_numpy(r, 1, s, e), want)


@pytest.mark.skipif(not os.path.isdir(REF_TEST), reason="reference fixtures only exist in the build container")
def test_golden_streams_match_reference_bams():
    from oracle import bamio
    for name, key in (("t", "t"), ("hla", "hla"), ("t-empty", "t_empty")):
        _, contigs, reads, total = bamio.read_bam(os.path.join(REF_TEST, name + ".bam"))
        gc, gr, z = H.load_golden_bam(key)
        assert contigs == gc and int(z["n_records_total"]) == total
        assert set(reads) == set(gr)
        for tid in reads:
            for f in ("pos", "flag", "mapq", "cigar_off", "cigar"):
                assert np.array_equal(getattr(reads[tid], f), getattr(gr[tid], f))
