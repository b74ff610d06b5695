"""-m gpu: the device-side BAM read (gd_inflate_bgzf, gd_ingest_bgzf) against the host-side
truth: zlib for the inflate, the record streams the BAM was written from for the decode.
Replaces, on the device, the BGZF inflate + record decode every `samtools depth` child of
the reference performs (depth/depth.go:45)."""
import struct
import zlib

import numpy as np
import pytest

from oracle import bamio, pyoracle as po
from tests import helpers as H

pytestmark = pytest.mark.gpu

OPT_INFLATE_KERNEL = 20
# GD_OPT_INFLATE_KERNEL: 0 a lane per member (the default), 1 a workgroup per member first (gd_inflate_wave.hpp) -- every inflate
# test runs on either and must see zlib's bytes and the same status words
KERNELS = pytest.mark.parametrize("kernel", [0, 1], ids=["lane-per-member", "workgroup-per-member"])


@KERNELS
@pytest.mark.parametrize("level", [0, 1, 6, 9])
def test_inflate_equals_zlib(tmp_path, level, kernel):
    from goleft_amd.engine import DepthEngine
    contigs, reads, _ = H.load_golden_bam("t")
    p = str(tmp_path / "a.bam")
    bamio.write_bam(p, contigs, reads, unplaced=2, level=level)      # level 0: stored blocks; the EOF member: a fixed block
    data = open(p, "rb").read()
    with DepthEngine(0) as eng:
        eng.set_option(OPT_INFLATE_KERNEL, kernel)
        got, status = eng.inflate_bgzf(data)
    assert (status == 0).all()
    assert got == bamio.bgzf_decompress(data)


@KERNELS
def test_inflate_varied_payloads(kernel):
    from goleft_amd.engine import DepthEngine
    rng = np.random.default_rng(5)
    parts = [b"", b"A", b"ACGT" * 5000, rng.integers(0, 256, 60000, dtype=np.uint8).tobytes(),
             bytes(65280), (b"x" * 300 + bytes(range(256))) * 100, rng.integers(0, 4, 65000, dtype=np.uint8).tobytes()]
    for level in (1, 9):
        data = b"".join(bamio.bgzf_compress(x, level=level) for x in parts)
        with DepthEngine(0) as eng:
            eng.set_option(OPT_INFLATE_KERNEL, kernel)
            got, status = eng.inflate_bgzf(data)
        assert (status == 0).all()
        assert got == b"".join(parts)


@KERNELS
def test_inflate_crc_at_slice_boundaries(kernel):
    """The CRC kernel gives a wave to each member and cuts it into 1 KB slices aligned to the member's end: lengths around
    every boundary, each once with its own trailer (status 0) and once with one bit of the trailer's CRC flipped (18)."""
    from goleft_amd.engine import DepthEngine
    rng = np.random.default_rng(5)
    sizes = (1, 7, 8, 9, 1023, 1024, 1025, 2047, 2048, 2049, 3071, 4096, 32767, 32768, 65279, 65280)
    parts = [rng.integers(0, 256, n, dtype=np.uint8).tobytes() for n in sizes]
    members = []
    for x in parts:
        first = []
        m = bamio.bgzf_compress(x, level=1 if len(x) < 40000 else 0, sizes=first)
        assert len(first) == 1
        members.append(m[:first[0]])                                                               # (without the EOF member)
    assert all(bamio.bgzf_decompress(m) == x for m, x in zip(members, parts))
    damaged = []
    for i, m in enumerate(members):
        b = bytearray(m)
        b[len(b) - 8 + (i % 4)] ^= 1 << (i % 8)                                                    # the trailer's CRC32
        damaged.append(bytes(b))
    with DepthEngine(0) as eng:
        eng.set_option(OPT_INFLATE_KERNEL, kernel)
        got, status = eng.inflate_bgzf(b"".join(members))
        assert (status == 0).all() and got == b"".join(parts)
        got, status = eng.inflate_bgzf(b"".join(damaged))
        assert (status == 18).all() and got == b"".join(parts)
        mixed = b"".join(d if i % 2 else m for i, (m, d) in enumerate(zip(members, damaged)))
        got, status = eng.inflate_bgzf(mixed)
        assert [int(v) for v in status] == [18 if i % 2 else 0 for i in range(len(sizes))] and got == b"".join(parts)


@KERNELS
@pytest.mark.parametrize("seed", range(4))
def test_inflate_random_streams(seed, kernel):
    # many members of random size / entropy / compression level and strategy in one call,
    # incl. Z_FIXED (fixed Huffman blocks), Z_RLE and Z_HUFFMAN_ONLY streams
    from goleft_amd.engine import DepthEngine
    rng = np.random.default_rng(seed)
    parts, blobs = [], []
    for _ in range(150):
        n = int(rng.integers(0, 65281))
        kind = rng.integers(0, 5)
        if kind == 0:
            x = rng.integers(0, 256, n, dtype=np.uint8)
        elif kind == 1:
            x = rng.integers(0, 4, n, dtype=np.uint8) + 65
        elif kind == 2:
            x = np.repeat(rng.integers(0, 256, n // 50 + 1, dtype=np.uint8), 50)[:n]
        elif kind == 3:
            base = rng.integers(0, 256, 97, dtype=np.uint8)
            x = np.tile(base, n // 97 + 1)[:n]
        else:
            x = np.zeros(n, np.uint8)
        x = x.tobytes()
        level = int(rng.integers(0, 10))
        strategy = int(rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_RLE, zlib.Z_HUFFMAN_ONLY, zlib.Z_FILTERED]))
        co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
        c = co.compress(x) + co.flush()
        if len(c) + 26 > 65536:                            # would not fit a BGZF member: store a smaller piece
            x = x[:30000]
            co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
            c = co.compress(x) + co.flush()
        blobs.append(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(c) + 25) + c
                     + struct.pack("<II", zlib.crc32(x) & 0xFFFFFFFF, len(x)))
        parts.append(x)
    with DepthEngine(0) as eng:
        eng.set_option(OPT_INFLATE_KERNEL, kernel)
        got, status = eng.inflate_bgzf(b"".join(blobs))
    assert (status == 0).all()
    assert got == b"".join(parts)


@KERNELS
def test_inflate_reports_corrupt_members(kernel):
    from goleft_amd.engine import DepthEngine
    good = bamio.bgzf_compress(b"hello world, " * 1000, level=6)
    bad = bytearray(good)
    bad[40] ^= 0x55                                        # inside the deflate payload of the first member
    with DepthEngine(0) as eng:
        eng.set_option(OPT_INFLATE_KERNEL, kernel)
        got, status = eng.inflate_bgzf(bytes(bad))
        assert status[0] != 0                              # a broken stream or, if it still decodes, its CRC32
        # a payload that inflates cleanly but to other bytes: only the CRC can tell
        x = b"hello world, " * 1000
        y = b"hello w0rld, " * 1000
        cy = zlib.compressobj(6, zlib.DEFLATED, -15)
        c = cy.compress(y) + cy.flush()
        member = (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(c) + 25) + c
                  + struct.pack("<II", zlib.crc32(x) & 0xFFFFFFFF, len(x)))
        got, status = eng.inflate_bgzf(member)
        assert status[0] == 18 and got == y
        got, status = eng.inflate_bgzf(member, check_crc=False)
        assert status[0] == 0


def ingest_contig(eng, path, tid):
    data = open(path, "rb").read()
    anchors = bamio.read_bai_linear(path + ".bai")[tid]
    if len(anchors) == 0:
        return 0
    return eng.ingest_bgzf(tid, data, 0, anchors)


@KERNELS
@pytest.mark.parametrize("name", ["t", "hla"])
def test_ingest_fixture_bam(tmp_path, name, kernel):
    from goleft_amd.engine import DepthEngine
    contigs, reads, _ = H.load_golden_bam(name)
    p = str(tmp_path / "a.bam")
    bamio.write_bam(p, contigs, reads, unplaced=3, index=True)
    with DepthEngine(0) as eng:
        eng.set_option(OPT_INFLATE_KERNEL, kernel)
        eng.set_params(window_size=100, min_mapq=1, min_cov=4)
        eng.set_contigs([c[1] for c in contigs])
        n = {tid: ingest_contig(eng, p, tid) for tid in range(len(contigs))}
        eng.compute()
        for tid, (_, L) in enumerate(contigs):
            r = reads.get(tid, H.empty_reads())
            assert n[tid] == r.n
            assert np.array_equal(eng.perbase(tid), po.perbase_c(r, 1, 0, L))


def test_ingest_several_references_in_one_pass(tmp_path):
    """gd_ingest_decode: one fed byte range (the whole file), one decode per reference -- many small
    contigs, one without records, tiny pieces; then the range is dropped and a second pass works."""
    from goleft_amd.engine import DepthEngine, GdError
    rng = np.random.default_rng(33)
    lens = [40_000, 900, 120_000, 5_000, 2_500, 64_000, 1]
    contigs = [("s%d" % i, l) for i, l in enumerate(lens)]
    reads = {t: H.random_reads(rng, l, int(rng.integers(1, 3000)), max_len=100)
             for t, l in enumerate(lens) if t not in (3, 6)}
    p = str(tmp_path / "g.bam")
    bamio.write_bam(p, contigs, reads, unplaced=4, index=True)
    data = open(p, "rb").read()
    lin = bamio.read_bai_linear(p + ".bai")
    with DepthEngine(0) as eng:
        eng.set_params(window_size=100, min_mapq=1, min_cov=4)
        eng.set_contigs(lens)
        refs = [(t, t, lin[t]) for t in range(len(lens)) if len(lin[t])]
        assert [t for t, _, _ in refs] == sorted(reads)
        counts = eng.ingest_bgzf_refs(data, 0, refs, piece=50_001)
        assert counts == [reads[t].n for t, _, _ in refs]
        eng.compute()
        for t, l in enumerate(lens):
            r = reads.get(t, H.empty_reads())
            assert np.array_equal(eng.perbase(t), po.perbase_c(r, 1, 0, l))
        # a second pass over a sub-range (from reference 2 on) replaces those contigs' records
        beg = int(lin[2][0]) >> 16
        sub = [(t, t, lin[t]) for t in (2, 4, 5)]
        assert eng.ingest_bgzf_refs(data[beg:], beg, sub) == [reads[t].n for t in (2, 4, 5)]
        eng.compute()
        assert np.array_equal(eng.perbase(5), po.perbase_c(reads[5], 1, 0, lens[5]))
        # an anchor outside every member is refused and drops the range; the next read starts clean
        bogus = np.asarray([int(lin[2][0]) + (7 << 16)], np.uint64)
        with pytest.raises(GdError):
            eng.ingest_bgzf_refs(data, 0, [(0, 0, lin[0]), (2, 2, bogus)])
        assert eng.ingest_bgzf(0, data, 0, lin[0]) == reads[0].n
        # anchors of a LATER reference decoded as an earlier one: its walk ends at the first record (no records, no
        # error: a sorted BAM goes on with later references) ...
        assert eng.ingest_bgzf_refs(data, 0, [(1, 1, lin[2])]) == [0]
        # ... anchors of an EARLIER reference: records a sorted BAM cannot hold after this contig's -- refused
        with pytest.raises(GdError):
            eng.ingest_bgzf_refs(data, 0, [(1, 2, lin[1])])
        assert eng.ingest_bgzf(0, data, 0, lin[0]) == reads[0].n


def test_ingest_two_ranges_pending(tmp_path):
    """The next ranges are fed before the previous one is decoded (what the CLI does so that the inflate
    tail of one pass overlaps the next upload; three may be pending since round 4): decode / release act on
    the oldest, a fourth begin is refused."""
    from goleft_amd.engine import DepthEngine, GdError
    rng = np.random.default_rng(44)
    lens = [90_000, 30_000, 150_000, 700]
    contigs = [("q%d" % i, l) for i, l in enumerate(lens)]
    reads = {t: H.random_reads(rng, l, int(rng.integers(500, 6000)), max_len=100) for t, l in enumerate(lens)}
    p = str(tmp_path / "q.bam")
    bamio.write_bam(p, contigs, reads, unplaced=2, index=True)
    data = open(p, "rb").read()
    lin = bamio.read_bai_linear(p + ".bai")
    start = [int(lin[t][0]) >> 16 for t in range(4)]
    with DepthEngine(0) as eng:
        eng.set_params(window_size=100, min_mapq=1, min_cov=4)
        eng.set_contigs(lens)
        eng.ingest_feed_range(data[:start[2] + 65536 + 26], 0, piece=40_000)       # references 0 and 1
        eng.ingest_feed_range(data[start[2]:start[3] + 65536 + 26], start[2], piece=1 << 20)   # reference 2
        eng.ingest_feed_range(data[start[3]:], start[3], piece=1 << 20)            # reference 3
        with pytest.raises(GdError):
            eng.ingest_feed_range(data, 0)                                         # a fourth pending range: refused,
        assert eng.ingest_decode(0, 0, lin[0]) == reads[0].n                       # nothing changed
        assert eng.ingest_decode(1, 1, lin[1]) == reads[1].n
        eng.ingest_release()
        assert eng.ingest_decode(2, 2, lin[2]) == reads[2].n
        eng.ingest_release()
        assert eng.ingest_decode(3, 3, lin[3]) == reads[3].n
        eng.ingest_release()
        with pytest.raises(GdError):
            eng.ingest_release()                                                   # nothing pending
        eng.compute()
        for t, l in enumerate(lens):
            assert np.array_equal(eng.perbase(t), po.perbase_c(reads[t], 1, 0, l))
        # a fed range nobody decodes is dropped by the one-shot form
        eng.ingest_feed_range(data[:start[2] + 65536 + 26], 0)
        assert eng.ingest_bgzf(3, data[start[3]:], start[3], lin[3]) == reads[3].n


def test_ingest_many_anchors_and_long_cigars(tmp_path):
    # 43 anchors (700 kb contig), reads with up to 70000 ops (stored through the CG:B,I tag)
    from goleft_amd.engine import DepthEngine
    rng = np.random.default_rng(21)
    L1, L2 = 700_000, 150_000
    r1 = H.random_reads(rng, L1, 60_000, max_len=120)
    r2 = H.long_cigar_reads(rng, L2, [70000, 5, 66000, 300, 1, 65536], max_step=3)
    contigs = [("c1", L1), ("empty", 5000), ("c2", L2)]
    p = str(tmp_path / "m.bam")
    bamio.write_bam(p, contigs, {0: r1, 2: r2}, unplaced=5, index=True)
    lin = bamio.read_bai_linear(p + ".bai")
    assert len(lin[0]) > 30 and len(lin[1]) == 0
    with DepthEngine(0) as eng:
        eng.set_params(window_size=250, min_mapq=0, min_cov=4)
        eng.set_contigs([c[1] for c in contigs])
        assert ingest_contig(eng, p, 0) == r1.n
        assert ingest_contig(eng, p, 2) == r2.n
        eng.compute()
        assert np.array_equal(eng.perbase(0), po.perbase_c(r1, 0, 0, L1))
        assert np.array_equal(eng.perbase(2), po.perbase_c(r2, 0, 0, L2))
        assert eng.perbase(1).sum() == 0


@pytest.mark.parametrize("piece", [1 << 30, 70_001, 4096, 333])
def test_ingest_streamed_in_pieces(tmp_path, piece):
    # the same read fed in pieces: members complete at arbitrary feed boundaries
    from goleft_amd.engine import DepthEngine
    contigs, reads, _ = H.load_golden_bam("t")
    p = str(tmp_path / "a.bam")
    bamio.write_bam(p, contigs, reads, unplaced=3, index=True)
    data = open(p, "rb").read()
    lin = bamio.read_bai_linear(p + ".bai")
    with DepthEngine(0) as eng:
        eng.set_params(window_size=100, min_mapq=1, min_cov=4)
        eng.set_contigs([c[1] for c in contigs])
        for tid in (0, 1):
            assert eng.ingest_bgzf_stream(tid, data, 0, lin[tid], piece) == reads[tid].n
        eng.compute()
        for tid, (_, L) in enumerate(contigs):
            assert np.array_equal(eng.perbase(tid), po.perbase_c(reads[tid], 1, 0, L))


def test_ingest_rejects_stale_anchor_and_corrupt_data(tmp_path):
    from goleft_amd.engine import DepthEngine, GdError
    contigs, reads, _ = H.load_golden_bam("t")
    p = str(tmp_path / "a.bam")
    bamio.write_bam(p, contigs, reads, index=True)
    data = open(p, "rb").read()
    anchors = bamio.read_bai_linear(p + ".bai")[1]
    with DepthEngine(0) as eng:
        eng.set_contigs([c[1] for c in contigs])
        with pytest.raises(GdError):                        # not a record start
            eng.ingest_bgzf(1, data, 0, np.array([anchors[0], anchors[1] + 7], np.uint64))
        with pytest.raises(GdError):                        # not inside any member
            eng.ingest_bgzf(1, data, 0, np.array([anchors[0] + (123 << 16)], np.uint64))
        bad = bytearray(data)
        bad[len(bad) // 2] ^= 0xff
        with pytest.raises(GdError):
            eng.ingest_bgzf(1, bytes(bad), 0, anchors)
        assert eng.ingest_bgzf(1, data, 0, anchors) == reads[1].n     # the context is still usable


def test_ingest_state_errors_and_pinned_alloc(tmp_path):
    import ctypes as C
    from goleft_amd import _lib
    from goleft_amd.engine import DepthEngine, GdError
    lib = _lib.load()
    contigs, reads, _ = H.load_golden_bam("t")
    p = str(tmp_path / "a.bam")
    bamio.write_bam(p, contigs, reads, index=True)
    data = open(p, "rb").read()
    anchors = bamio.read_bai_linear(p + ".bai")[0]
    with DepthEngine(0) as eng:
        eng.set_contigs([c[1] for c in contigs])
        raw = np.frombuffer(data, np.uint8)
        assert lib.gd_ingest_feed(eng._ctx, raw.ctypes.data, 10) == -4            # GD_E_STATE: no begin
        cnt = C.c_uint64()
        a = np.ascontiguousarray(anchors, np.uint64)
        assert lib.gd_ingest_finish(eng._ctx, 0, 0, a.ctypes.data, a.size, C.byref(cnt)) == -4
        # begin, feed only half, finish -> refused, and the context recovers
        nm = C.c_size_t()
        lib.gd_bgzf_members(raw.ctypes.data, raw.size, 0, None, None, None, None, None, C.byref(nm))
        n = nm.value
        moff, msize, mhdr = np.zeros(n, np.uint64), np.zeros(n, np.uint32), np.zeros(n, np.uint16)
        misz, mcrc = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        assert lib.gd_bgzf_members(raw.ctypes.data, raw.size, n, moff.ctypes.data, msize.ctypes.data, mhdr.ctypes.data,
                                   misz.ctypes.data, mcrc.ctypes.data, C.byref(nm)) == 0
        used = int(moff[-1]) + int(msize[-1])
        assert used == raw.size
        assert lib.gd_ingest_begin(eng._ctx, used, 0, n, moff.ctypes.data, msize.ctypes.data, mhdr.ctypes.data,
                                   misz.ctypes.data, mcrc.ctypes.data) == 0
        assert lib.gd_ingest_feed(eng._ctx, raw.ctypes.data, used // 2) == 0
        assert lib.gd_ingest_finish(eng._ctx, 0, 0, a.ctypes.data, a.size, C.byref(cnt)) == -4
        assert lib.gd_ingest_feed(eng._ctx, raw.ctypes.data, 10) == -4            # finish released the read
        assert eng.ingest_bgzf(0, data, 0, anchors) == reads[0].n
        # page-locked staging for callers that read the file themselves
        ptr = C.c_void_p()
        assert lib.gd_host_alloc(eng._ctx, 1 << 20, C.byref(ptr)) == 0 and ptr.value
        C.memset(ptr, 7, 1 << 20)
        assert lib.gd_host_free(eng._ctx, ptr) == 0


def _record_starts(inflated: bytes):
    """Offsets of the records of a BAM's inflated stream (after the header and reference table)."""
    l_text, = struct.unpack_from("<i", inflated, 4)
    p = 8 + l_text
    n_ref, = struct.unpack_from("<i", inflated, p)
    p += 4
    for _ in range(n_ref):
        l_name, = struct.unpack_from("<i", inflated, p)
        p += 8 + l_name
    out = []
    while p + 4 <= len(inflated):
        bs, = struct.unpack_from("<i", inflated, p)
        out.append(p)
        p += 4 + bs
    return out


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("GOLEFT_FUZZ_SEEDS", "40"))))
def test_device_bam_walker_mutation_fuzz(tmp_path, seed):
    """Random byte flips in the INFLATED record stream of a BAM (re-compressed, so every BGZF CRC holds and only the
    record walker can notice), biased to the fields a walker trusts -- block_size, refID, pos, l_read_name,
    n_cigar_op, l_seq, the CIGAR -- and flips in the compressed members: the device read must either refuse the file
    (GD_E_INVALID / GD_E_UNSORTED) or deliver exactly what the host decoder delivers; the context stays usable and
    the GPU never faults (a fault would end the test process).  The reference gets this from samtools
    (depth/depth.go:395-399 just reports the child's exit code)."""
    from goleft_amd import _hostlib
    from goleft_amd.engine import DepthEngine, GdError
    contigs, reads, _ = H.load_golden_bam("t")
    p0 = str(tmp_path / "a.bam")
    bamio.write_bam(p0, contigs, reads, unplaced=3, index=True)
    data0 = open(p0, "rb").read()
    infl = bytearray(bamio.bgzf_decompress(data0))
    lin0 = bamio.read_bai_linear(p0 + ".bai")
    # anchors as uncompressed offsets (every data member of bamio's files holds `block` bytes)
    block = 0xff00
    sizes0: list = []
    bamio.bgzf_compress(bytes(infl), level=1, sizes=sizes0)
    coff0 = np.concatenate([[0], np.cumsum(sizes0)]).astype(np.int64)
    def to_uncompressed(v):
        k = int(np.searchsorted(coff0, int(v) >> 16))
        return k * block + (int(v) & 0xffff)
    anchors_u = {t: [to_uncompressed(v) for v in lin0[t]] for t in lin0}
    starts = _record_starts(bytes(infl))
    rng = np.random.default_rng(1000 + seed)
    mode = seed % 4
    if mode < 3:
        # mode 0: one or two flips in fields whose every value is a legal record (pos, flag / mapq, CIGAR bytes): most
        # of these files stay readable and are COMPARED; mode 1: one flip anywhere in the fixed fields; mode 2: up to five
        for _ in range(int(rng.integers(1, 3 if mode < 2 else 6))):
            s = starts[int(rng.integers(0, len(starts)))]
            field = int(rng.choice([2, 6, 7])) if mode == 0 else int(rng.integers(0, 9))
            # block_size, refID, pos, l_read_name, n_cigar_op, l_seq, flag / mapq, first CIGAR bytes, anywhere in the record
            at = s + [int(rng.integers(0, 4)), 4 + int(rng.integers(0, 4)), 8 + int(rng.integers(0, 4)), 12, 16 + int(rng.integers(0, 2)),
                      20 + int(rng.integers(0, 4)), 13 + int(rng.integers(0, 6)), 36 + int(rng.integers(0, 24)),
                      int(rng.integers(0, 200))][field]
            if at < len(infl):
                infl[at] = (infl[at] ^ (1 << int(rng.integers(0, 8)))) if rng.random() < 0.6 else int(rng.integers(0, 256))
    sizes: list = []
    data = bytearray(bamio.bgzf_compress(bytes(infl), level=1, sizes=sizes))
    if mode == 3:                                            # the compressed bytes themselves
        for _ in range(int(rng.integers(1, 4))):
            data[int(rng.integers(0, len(data) - 28))] ^= 1 << int(rng.integers(0, 8))
    data = bytes(data)
    coff = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    p1 = str(tmp_path / "m.bam")
    open(p1, "wb").write(data)
    # what the host decoder makes of it
    host = None
    try:
        _, host, _ = _hostlib.read_bam(p1, threads=2)
        for t, (pos, flag, mapq, off, cig) in host.items():
            if t >= len(contigs) or (len(pos) > 1 and (np.diff(pos) < 0).any()):
                host = None                                  # a reference out of range / unsorted: gd_commit refuses it
                break
    except OSError:
        host = None
    with DepthEngine(0) as eng:
        from goleft_amd.engine import OPT_BAM_REFS
        eng.set_params(window_size=100, min_mapq=1, min_cov=4)
        eng.set_contigs([c[1] for c in contigs])
        eng.set_option(OPT_BAM_REFS, len(contigs))
        got = {}
        try:
            for t in sorted(anchors_u):
                if len(anchors_u[t]) == 0:
                    continue
                anchors = np.array([(int(coff[o // block]) << 16) | (o % block) for o in anchors_u[t]], np.uint64)
                got[t] = eng.ingest_bgzf(t, data, 0, anchors)
            eng.compute()
            dev = {t: eng.perbase(t) for t in got}
        except GdError as e:
            assert e.status in (-1, -7, -5), e               # GD_E_INVALID, GD_E_UNSORTED (GD_E_RANGE: a length past the limits)
            dev = None
        if dev is not None:
            assert host is not None, "the device read accepted a file the host decoder refuses"
            for t in got:
                r = host.get(t)
                want = po.perbase_c(po.Reads(*r), 1, 0, contigs[t][1]) if r is not None else np.zeros(contigs[t][1], np.int32)
                assert got[t] == (len(r[0]) if r is not None else 0), (t, got[t])
                assert np.array_equal(dev[t], want), t
        # the context is still usable: the unmodified file goes through
        for t in sorted(lin0):
            if len(lin0[t]):
                assert eng.ingest_bgzf(t, data0, 0, lin0[t]) == reads[t].n
        eng.compute()
        for t in reads:
            assert np.array_equal(eng.perbase(t), po.perbase_c(reads[t], 1, 0, contigs[t][1]))


def test_ingest_a_reference_in_parts(tmp_path):
    """gd_ingest_decode_part: a reference read as several byte ranges cut at its .bai anchors -- part k holds the bytes
    from the member of its first anchor through the member of part k + 1's first anchor, its records are appended to
    the contig's arrays, coordinate order is checked ACROSS parts, and what each call refuses."""
    from goleft_amd.engine import DepthEngine, GdError
    rng = np.random.default_rng(71)
    lens = [900_000, 50_000]
    contigs = [("p%d" % i, l) for i, l in enumerate(lens)]
    reads = {0: H.random_reads(rng, lens[0], 40_000, max_ops=4, max_len=90), 1: H.random_reads(rng, lens[1], 3_000, max_len=90)}
    p = str(tmp_path / "p.bam")
    bamio.write_bam(p, contigs, reads, unplaced=2, index=True)
    data = open(p, "rb").read()
    lin = bamio.read_bai_linear(p + ".bai")
    an = np.asarray(lin[0], np.uint64)
    assert len(an) > 12
    want = po.perbase_c(reads[0], 1, 0, lens[0])
    end_of_0 = (int(lin[1][0]) >> 16) + 65536 + 26                  # reference 0's records end in reference 1's first member
    with DepthEngine(0) as eng:
        eng.set_params(window_size=100, min_mapq=1, min_cov=4)
        eng.set_contigs(lens)
        for cuts in ([0, 5, len(an)], [0, 1, 2, 7, len(an) - 1, len(an)], list(range(len(an) + 1))):
            # (cuts only where the next part starts in a LATER member, as the CLI's planner does)
            keep = [0]
            for c in cuts[1:]:
                if c == len(an) or (int(an[c]) >> 16) > (int(an[keep[-1]]) >> 16):
                    keep.append(c)
            cuts = keep if keep[-1] == len(an) else keep + [len(an)]
            total = 0
            for k in range(len(cuts) - 1):
                lo, hi = cuts[k], cuts[k + 1]
                beg = int(an[lo]) >> 16
                end = min(len(data), (int(an[hi]) >> 16) + 65536 + 26) if hi < len(an) else min(len(data), end_of_0)
                eng.ingest_feed_range(data[beg:end], beg, piece=70_001)
                total += eng.ingest_decode_part(0, 0, an[lo:hi], int(an[hi]) if hi < len(an) else 0, append=k > 0, release=True,
                                                expect_scale=0.0 if k else 3.0)
            assert total == reads[0].n, cuts
            eng.compute()
            assert np.array_equal(eng.perbase(0), want), cuts
            assert eng.stats().reruns == 0
        # the second part fed again: records that go back in coordinates -> refused (the arrays keep what they held)
        lo, hi = 5, len(an)
        beg = int(an[lo]) >> 16
        eng.ingest_feed_range(data[beg:end_of_0], beg)
        with pytest.raises(GdError) as ei:
            eng.ingest_decode_part(0, 0, an[lo:hi], 0, append=True, release=True)
        assert ei.value.status == -7
        # an end anchor outside the fed range / not behind the last anchor
        beg = int(an[0]) >> 16
        eng.ingest_feed_range(data[beg:(int(an[5]) >> 16) + 65536 + 26], beg)
        with pytest.raises(GdError) as ei:
            eng.ingest_decode_part(0, 0, an[0:5], int(an[-1]), release=True)
        assert ei.value.status == -1
        eng.ingest_feed_range(data[beg:(int(an[5]) >> 16) + 65536 + 26], beg)
        with pytest.raises(GdError) as ei:
            eng.ingest_decode_part(0, 0, an[0:5], int(an[2]), release=True)
        assert ei.value.status == -1
        # a reference that ENDED in an earlier part must not resume: part 1 = reference 0's tail (a record of reference 1
        # ends the walk), then reference 0's first part again as an appended part
        eng.ingest_feed_range(data[int(an[5]) >> 16:end_of_0], int(an[5]) >> 16)
        eng.ingest_decode_part(0, 0, an[5:], 0, release=True)
        eng.ingest_feed_range(data[beg:(int(an[5]) >> 16) + 65536 + 26], beg)
        with pytest.raises(GdError) as ei:
            eng.ingest_decode_part(0, 0, an[0:5], int(an[5]), append=True, release=True)
        assert ei.value.status == -7
        # and the context reads the whole file as before
        assert eng.ingest_bgzf(0, data, 0, lin[0]) == reads[0].n
        eng.compute()
        assert np.array_equal(eng.perbase(0), want)


@KERNELS
def test_inflate_a_member_of_fifty_thousand_empty_blocks(kernel):
    """a VALID member may consist of tens of thousands of empty blocks (10 bits each with the fixed
    code), each of which waits up to 32 iterations of the symbol loop for the other lanes' headers -- more iterations
    than the loop's former backstop of 2^20 allowed, which refused the member (err 19) although zlib inflates it."""
    import struct
    from goleft_amd.engine import DepthEngine
    n_empty = 50_000
    payload = b"the bytes behind fifty thousand empty blocks"
    bits = []
    for _ in range(n_empty):
        bits += [0, 1, 0] + [0] * 7                          # BFINAL 0, BTYPE 01 (fixed), end-of-block (0000000)
    bits += [1, 0, 0]                                        # the last block: stored
    while len(bits) % 8:
        bits.append(0)
    raw = bytearray()
    for i in range(0, len(bits), 8):
        raw.append(sum(b << k for k, b in enumerate(bits[i:i + 8])))
    raw += struct.pack("<HH", len(payload), len(payload) ^ 0xffff) + payload
    assert zlib.decompress(bytes(raw), -15) == payload       # the yardstick accepts it
    assert len(raw) + 26 < 65536
    member = (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(raw) + 25) + bytes(raw) +
              struct.pack("<II", zlib.crc32(payload) & 0xffffffff, len(payload)))
    data = member + bamio.bgzf_compress(b"ACGT" * 1000)      # (another member in the same wave, and the EOF marker)
    with DepthEngine(0) as eng:
        eng.set_option(OPT_INFLATE_KERNEL, kernel)
        got, status = eng.inflate_bgzf(data)
    assert (status == 0).all(), status
    assert got == payload + b"ACGT" * 1000


def test_measurement_switches_are_refused_by_a_release_library():
    """include/goleft_depth.h lists the switches of measurement builds (-DGD_MEASURE) apart: the library that ships keeps them at
    their defaults, accepts the default and answers GD_E_INVALID to anything else (a cgo host could set a
    switch that was only ever meant for a measurement)."""
    from goleft_amd.engine import DepthEngine, GdError
    with DepthEngine(0) as eng:
        for opt, v in ((15, 2), (16, 4096), (17, 1), (21, 4), (22, 1), (13, 3)):
            with pytest.raises(GdError) as ei:
                eng.set_option(opt, v)
            assert ei.value.status == -1, (opt, v, ei.value.status)
        for opt, v in ((15, 1), (16, 0), (17, 0), (21, 8), (22, 0), (13, 0), (13, 1)):
            eng.set_option(opt, v)
