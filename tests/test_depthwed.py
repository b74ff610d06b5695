"""depthwed (BASELINE.json config 4): the shared %.4g rounding step, the oracle
restatement of depthwed/depthwed.go and the C++ host twin (CPU only)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import pyoracle as po
from tests import helpers as H


def cells_c(sums, lens):
    from goleft_amd import _hostlib
    lib = _hostlib.load()
    sums = np.ascontiguousarray(sums, np.int64)
    lens = np.ascontiguousarray(lens, np.int64)
    out = np.empty(len(sums), np.int64)
    lib.gdh_depthwed_cells(sums.ctypes.data, lens.ctypes.data, len(sums), out.ctypes.data)
    return out


def cells_py(sums, lens):
    """What the reference chain computes: goleft depth prints "%.4g" of
    float64(sum)/float64(len) (depth/depth.go:188,:301); depthwed parses it back
    and takes int(0.5 + x) (depthwed/depthwed.go:96,:103)."""
    out = np.empty(len(sums), np.int64)
    for i, (s, l) in enumerate(zip(sums.tolist(), lens.tolist())):
        mean = 0.0 if s == 0 else float(s) / float(l)
        out[i] = po.depthwed_cell("%.4g" % mean)
    return out


def test_cell_known_values():
    # (sum, len) -> cell; ties of the 4-digit rounding go to even, then .5 rounds up
    cases = [(0, 250, 0), (1, 250, 0), (124, 250, 0), (125, 250, 1), (1001000, 1000, 1001),
             (12345, 10, 1234),      # 1234.5 -> "1234" (tie to even)
             (12355, 10, 1236),      # 1235.5 -> "1236"
             (123450, 1000, 124),    # the double nearest 123.45 lies ABOVE the tie -> "123.5" -> 124
             (99996, 10, 10000),     # 9999.6 -> "1e+04"
             (45850, 1000, 46), (999, 2000, 0), (1000, 2000, 1), (3, 2, 2), (5, 2, 3)]
    s = np.array([c[0] for c in cases]); l = np.array([c[1] for c in cases])
    want = np.array([c[2] for c in cases])
    assert np.array_equal(cells_py(s, l), want)
    assert np.array_equal(cells_c(s, l), want)


@pytest.mark.parametrize("seed", range(4))
def test_cell_random_against_printf(seed):
    rng = np.random.default_rng(seed)
    n = 150000
    lens = rng.choice([1, 2, 3, 7, 13, 50, 55, 60, 71, 200, 250, 1000, 2001, 16384, 10 ** 6,
                       10 ** 8, 2 ** 31 - 1], size=n).astype(np.int64)
    scale = 10.0 ** rng.uniform(-3, 5, size=n)
    sums = np.floor(lens * scale * rng.random(n)).astype(np.int64)
    sums[rng.random(n) < 0.02] = 0
    assert np.array_equal(cells_c(sums, lens), cells_py(sums, lens))


def test_cell_ties_and_boundaries():
    """Means sitting exactly on (or one unit of the sum away from) a 4-digit
    rounding tie or a x.5 boundary, for every magnitude."""
    sums, lens = [], []
    for l in (1, 2, 4, 5, 8, 10, 20, 40, 125, 250, 1000, 2000, 16384):
        for e in range(-1, 7):
            for d in (1000, 1001, 1234, 1235, 4999, 5000, 5001, 9998, 9999):
                for half in (0, 1):
                    # mean ~ (d + half/2) * 10^(e-3)
                    num = (2 * d + half) * 10 ** max(e, 0) * l
                    den = 2 * 10 ** 3 * 10 ** max(-e, 0)
                    base = num // den
                    for delta in (-1, 0, 1):
                        if base + delta >= 0:
                            sums.append(base + delta)
                            lens.append(l)
    s = np.array(sums, np.int64); l = np.array(lens, np.int64)
    assert np.array_equal(cells_c(s, l), cells_py(s, l))


def test_name_from_file():
    assert po.depthwed_name("/x/y/sample1.depth.bed") == "sample1"
    assert po.depthwed_name("s.depth.bed.gz") == "s"
    assert po.depthwed_name("plain.txt") == "plain.txt"


def _fixture_beds(tmp_path, W):
    """Three 'samples' over the same windows: the golden t.bam depth.bed at W and two
    perturbed copies (same rows, different means)."""
    base = H.golden_beds()["t"]["wg_w%d" % W]["depth"]
    rows = [r.split("\t") for r in base.strip().split("\n")]
    texts = [base]
    for mul in (0.37, 2.6):
        texts.append("".join("%s\t%s\t%s\t%.4g\n" % (r[0], r[1], r[2], float(r[3]) * mul) for r in rows))
    paths = []
    for i, t in enumerate(texts):
        p = tmp_path / ("s%d.depth.bed" % i)
        p.write_text(t)
        paths.append(str(p))
    return texts, paths


@pytest.mark.parametrize("W,size", [(250, 1000), (250, 250), (250, 600), (1000, 5000), (1000, 1000)])
def test_host_cli_matches_oracle(tmp_path, W, size):
    from goleft_amd import _hostlib
    texts, paths = _fixture_beds(tmp_path, W)
    want = po.depthwed_py(texts, [po.depthwed_name(p) for p in paths], size)
    out = str(tmp_path / "m.txt")
    arr = (C.c_char_p * len(paths))(*[p.encode() for p in paths])
    rc = _hostlib.load().gdh_depthwed_run(size, arr, len(paths), out.encode())
    assert rc == 0
    assert open(out).read() == want
    # and through the executable, gz input included
    exe = os.path.join(os.path.dirname(_hostlib.SO_PATH), "goleft-depth")
    subprocess.check_call(["gzip", "-k", paths[1]])
    got = subprocess.run([exe, "depthwed", "-s", str(size), paths[0], paths[1] + ".gz", paths[2]],
                         check=True, capture_output=True).stdout.decode()
    names = [po.depthwed_name(paths[0]), po.depthwed_name(paths[1] + ".gz"), po.depthwed_name(paths[2])]
    assert got == po.depthwed_py(texts, names, size)


def test_host_cli_unequal_records_is_an_error(tmp_path):
    from goleft_amd import _hostlib
    a = tmp_path / "a.depth.bed"; b = tmp_path / "b.depth.bed"
    a.write_text("c\t0\t250\t1\nc\t250\t500\t2\n")
    b.write_text("c\t0\t250\t1\n")
    arr = (C.c_char_p * 2)(str(a).encode(), str(b).encode())
    rc = _hostlib.load().gdh_depthwed_run(250, arr, 2, str(tmp_path / "o").encode())
    assert rc != 0
    with pytest.raises(RuntimeError):
        po.depthwed_py([a.read_text(), b.read_text()], ["a", "b"], 250)


from hypothesis import given, settings, strategies as st  # noqa: E402


@settings(max_examples=150, deadline=None)
@given(st.integers(0, 2 ** 31))
def test_host_cli_survives_malformed_rows(seed):
    """Garbage in the *.depth.bed inputs (missing columns, text for numbers, huge values, empty files,
    binary bytes): gdh_depthwed_run returns an error or writes something, it never crashes."""
    import tempfile
    from goleft_amd import _hostlib
    rng = np.random.default_rng(seed)
    junk = [b"", b"\n", b"chr1\n", b"chr1\t0\n", b"chr1\t0\t250\n", b"chr1\tx\ty\tz\n", b"chr1\t0\t250\t1e400\n",
            b"chr1\t-5\t99999999999999999999\tnan\n", b"\t\t\t\n", b"\x00\xff\x1f\x8b\n", b"chr1\t250\t0\t3\n",
            b"c" * 5000 + b"\t0\t250\t1\n", b"chr2\t0\t250\t7.25\textra\tcols\n"]
    with tempfile.TemporaryDirectory() as td:
        paths = []
        for i in range(int(rng.integers(1, 4))):
            rows = []
            for k in range(int(rng.integers(0, 30))):
                if rng.random() < 0.3:
                    rows.append(junk[int(rng.integers(0, len(junk)))])
                else:
                    rows.append(b"chr1\t%d\t%d\t%.4g\n" % (k * 250, k * 250 + 250, rng.random() * 60))
            p = os.path.join(td, "s%d.depth.bed" % i)
            open(p, "wb").write(b"".join(rows))
            paths.append(p)
        arr = (C.c_char_p * len(paths))(*[p.encode() for p in paths])
        rc = _hostlib.load().gdh_depthwed_run(int(rng.choice([1, 250, 1000, 10 ** 9])), arr, len(paths),
                                              os.path.join(td, "o.txt").encode())
        assert isinstance(rc, int)


@pytest.mark.parametrize("L,W,size", [(100001, 250, 1000), (35250, 250, 600), (999, 100, 100), (64000, 1000, 16000),
                                      (4001, 250, 1000), (1000, 250, 1000), (1, 250, 1000)])
def test_array_restatement_equals_the_line_by_line_one(L, W, size):
    """pyoracle.depthwed_cells_contig (what the full-size GPU test of BASELINE config 4 compares with) against
    depthwed_py on the BED text of the same window sums."""
    rng = np.random.default_rng(L + W)
    nw = (L + W - 1) // W
    cols = [rng.integers(0, 90 * W, size=nw), rng.integers(0, 3, size=nw) * rng.integers(0, 5 * W, size=nw),
            np.full(nw, 30 * W + 125)]
    beds = []
    for sums in cols:
        rows = []
        for k in range(nw):
            s, e = k * W, min(L, (k + 1) * W)
            mean = 0.0 if sums[k] == 0 else float(sums[k]) / float(e - s)
            rows.append("chrQ\t%d\t%d\t%s" % (s, e, "%.4g" % mean))
        beds.append("\n".join(rows) + "\n")
    want = po.depthwed_py(beds, ["a", "b", "c"], size).strip().split("\n")[1:]
    got = [po.depthwed_cells_contig(sums, L, W, size) for sums in cols]
    assert all(len(g[0]) == len(want) for g in got)
    for k, line in enumerate(want):
        t = line.split("\t")
        assert int(t[1]) == got[0][1][k] and int(t[2]) == got[0][2][k]
        assert [int(x) for x in t[3:]] == [int(g[0][k]) for g in got], (k, line)
