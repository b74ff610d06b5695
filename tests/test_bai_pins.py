"""CPU: what the reference's own index files pin.  `depth/test/t.bam.bai` and
`depth/test/hla.bam.bai` were written by htslib (`samtools index`), so they are reference-held vectors about the RECORD
STREAM of those BAMs (SAMv1 section 5.2): per reference, the pseudo-bin 37450 holds the virtual offsets of its first
record and of the end of its last one and the counts n_mapped / n_unmapped; the linear index holds, per 16 kb window, the
smallest virtual offset of a record overlapping it.  Both decoders of this repository (the pure-Python reader of oracle/
and the C++ host reader that feeds the ring; the device decoder in tests/test_gpu_ref_fixtures.py) must deliver record
streams that reproduce every one of those numbers: record boundaries, refIDs, positions, FLAG bit 0x4 and the reference
length of every CIGAR (bam_endpos) are then the ones htslib saw."""
import os
import struct
import zlib

import numpy as np
import pytest

from oracle import bamio
from tests import helpers as H

REF = os.path.join(H.GOLDEN, "ref")


@pytest.fixture(scope="module")
def hostlib():
    from goleft_amd import _hostlib
    _hostlib.load()
    return _hostlib


def parse_bai(path):
    """-> per reference: dict(bins={bin: [(beg, end)]}, meta=(ref_beg, ref_end, n_mapped, n_unmapped) or None, lin=uint64[])"""
    d = open(path, "rb").read()
    assert d[:4] == b"BAI\x01"
    n_ref, = struct.unpack_from("<i", d, 4)
    p = 8
    out = []
    for _ in range(n_ref):
        n_bin, = struct.unpack_from("<i", d, p)
        p += 4
        bins, meta = {}, None
        for _ in range(n_bin):
            b, n_chunk = struct.unpack_from("<Ii", d, p)
            p += 8
            chunks = [struct.unpack_from("<QQ", d, p + 16 * k) for k in range(n_chunk)]
            p += 16 * n_chunk
            if b == 37450:
                assert n_chunk == 2                                # SAMv1 5.2: (ref_beg, ref_end), (n_mapped, n_unmapped)
                meta = (chunks[0][0], chunks[0][1], chunks[1][0], chunks[1][1])
            else:
                bins[b] = chunks
        n_intv, = struct.unpack_from("<i", d, p)
        p += 4
        lin = np.frombuffer(d, "<u8", n_intv, p).copy()
        p += 8 * n_intv
        out.append({"bins": bins, "meta": meta, "lin": lin})
    n_no_coor = struct.unpack_from("<Q", d, p)[0] if p + 8 <= len(d) else None
    return out, n_no_coor


def records_with_virtual_offsets(path):
    """Every record of a BAM as (refID, pos, end, flag, voffset, voffset_after): an independent walk of the BGZF
    members (zlib) and of the block_size chain; end = pos + reference length of the CIGAR, pos + 1 when that is 0."""
    raw = open(path, "rb").read()
    members = []                                                   # (coffset, inflated start, inflated length)
    off, total, parts = 0, 0, []
    while off < len(raw):
        xlen, = struct.unpack_from("<H", raw, off + 10)
        bsize = None
        q = off + 12
        while q < off + 12 + xlen:
            if raw[q] == 66 and raw[q + 1] == 67:
                bsize, = struct.unpack_from("<H", raw, q + 4)
            q += 4 + struct.unpack_from("<H", raw, q + 2)[0]
        data = zlib.decompress(raw[off + 12 + xlen:off + bsize + 1 - 8], -15)
        members.append((off, total, len(data)))
        parts.append(data)
        total += len(data)
        off += bsize + 1
    d = b"".join(parts)
    starts = np.asarray([m[1] for m in members], np.int64)

    def voff(o):
        # htslib's bgzf_tell: a position at the very end of a member is the START of the next one
        k = int(np.searchsorted(starts, o, "right")) - 1
        while k + 1 < len(members) and o == members[k][1] + members[k][2]:
            k += 1
            if members[k][2] or k + 1 == len(members):
                break
        return (members[k][0] << 16) | (o - members[k][1])

    l_text, = struct.unpack_from("<i", d, 4)
    p = 8 + l_text
    n_ref, = struct.unpack_from("<i", d, p)
    p += 4
    for _ in range(n_ref):
        l_name, = struct.unpack_from("<i", d, p)
        p += 8 + l_name
    recs = []
    while p < len(d):
        bs, ref, pos, l_name, mapq, _bin, n_cig, flag = struct.unpack_from("<iiiBBHHH", d, p)
        cig = np.frombuffer(d, "<u4", n_cig, p + 36 + l_name)
        rlen = int(sum(int(c >> 4) for c in cig if (int(c) & 15) in (0, 2, 3, 7, 8)))
        end = pos + (rlen if rlen > 0 and not (flag & 4) else 1)   # bam_endpos
        recs.append((ref, pos, end, flag, voff(p), voff(p + 4 + bs)))
        p += 4 + bs
    return recs, n_ref


@pytest.mark.parametrize("name", ["t", "hla"])
def test_the_reference_index_describes_the_stream_the_decoders_deliver(name, hostlib):
    bam = os.path.join(REF, name + ".bam")
    idx, n_no_coor = parse_bai(bam + ".bai")
    recs, n_ref = records_with_virtual_offsets(bam)
    assert len(idx) == n_ref
    _, _, py_reads, py_total = bamio.read_bam(bam)
    assert py_total == len(recs)
    _, host_reads, host_total = hostlib.read_bam(bam)
    assert host_total == len(recs)
    checked_meta = checked_lin = 0
    for ref in range(n_ref):
        mine = [r for r in recs if r[0] == ref]
        meta, lin = idx[ref]["meta"], idx[ref]["lin"]
        if not mine:
            assert meta is None or (meta[2] == 0 and meta[3] == 0)
            assert ref not in py_reads and ref not in host_reads
            continue
        # ---- pseudo-bin: where the reference's records begin and end, how many are mapped / unmapped ----
        assert meta is not None
        n_unmapped = sum(1 for r in mine if r[3] & 4)
        # (the file's LAST record: htslib closes the index with the reader's position at end of file -- behind the
        # empty EOF member -- hts_idx_finish(idx, bgzf_tell(fp)); every other reference ends where its last record ends)
        ref_end = mine[-1][5] if mine[-1] is not recs[-1] else os.path.getsize(bam) << 16
        assert meta == (mine[0][4], ref_end, len(mine) - n_unmapped, n_unmapped), (ref, meta)
        checked_meta += 1
        # ... and both decoders deliver exactly those records (a placed record has POS >= 0 in these files)
        for who, reads in (("python", py_reads), ("host", host_reads)):
            r = reads[ref]
            pos, flag = (r.pos, r.flag) if hasattr(r, "pos") else (r[0], r[1])
            assert len(pos) == meta[2] + meta[3], (who, ref)
            assert int((flag & 4 != 0).sum()) == meta[3], (who, ref)
            assert np.array_equal(pos, np.asarray([m[1] for m in mine], np.int32)), (who, ref)
        # ---- linear index: the smallest virtual offset of a record overlapping each 16 kb window ----
        want = {}
        for _, pos, end, _, v, _ in mine:
            for w in range(pos >> 14, ((end - 1) >> 14) + 1):
                want[w] = min(want.get(w, v), v)
        assert len(lin) == max(want) + 1, (ref, len(lin), max(want))
        for w, v in want.items():
            assert int(lin[w]) == v, (ref, w, int(lin[w]), v)
            checked_lin += 1
        # windows no record overlaps: htslib fills them from their right-hand neighbour (or leaves 0 in front)
        for w in range(len(lin)):
            if w not in want:
                assert int(lin[w]) in (0, int(lin[w + 1]) if w + 1 < len(lin) else 0), (ref, w)
    assert checked_meta >= 1 and checked_lin >= 1
    if n_no_coor is not None:
        assert n_no_coor == sum(1 for r in recs if r[0] < 0)


def test_the_committed_record_streams_are_those_records():
    """tests/golden/*_bam.npz (what every GPU parity test feeds the engine) against the same independent walk."""
    for name in ("t", "hla"):
        recs, _ = records_with_virtual_offsets(os.path.join(REF, name + ".bam"))
        contigs, reads, _ = H.load_golden_bam(name)
        for tid, r in reads.items():
            mine = [x for x in recs if x[0] == tid]
            assert np.array_equal(r.pos, np.asarray([m[1] for m in mine], np.int32))
            assert np.array_equal(r.flag, np.asarray([m[3] for m in mine], np.uint16))
            ends = H.ref_span(r) + r.pos
            want = np.asarray([m[2] for m in mine], np.int64)
            ok = (H.ref_span(r) > 0) & ((r.flag & 4) == 0)
            assert np.array_equal(ends[ok], want[ok])
