"""Minimal BGZF/BAM reader and writer in pure Python (TEST INFRASTRUCTURE ONLY).

Used to (a) decode the reference's fixture BAMs (/root/reference/depth/test/*.bam)
into the SoA record streams committed under tests/golden/, (b) write small BAM
files from record streams so that the product's C++ BAM reader can be
exercised on the GPU box, where /root/reference does not exist.

Format follows the SAM/BAM specification (SAMv1 section 4): BGZF = gzip members
with a 'BC' extra subfield, BAM = magic, header text, reference table, records.
"""
from __future__ import annotations

import struct
import zlib

import numpy as np

from .pyoracle import Reads


def bgzf_decompress(raw: bytes) -> bytes:
    out = []
    off = 0
    n = len(raw)
    while off < n:
        if raw[off:off + 4] != b"\x1f\x8b\x08\x04":
            raise ValueError("not a BGZF member at %d" % off)
        xlen, = struct.unpack_from("<H", raw, off + 10)
        p = off + 12
        bsize = None
        while p < off + 12 + xlen:
            si1, si2, slen = raw[p], raw[p + 1], struct.unpack_from("<H", raw, p + 2)[0]
            if si1 == 66 and si2 == 67 and slen == 2:
                bsize, = struct.unpack_from("<H", raw, p + 4)
            p += 4 + slen
        if bsize is None:
            raise ValueError("BGZF member without BC subfield")
        cdata = raw[off + 12 + xlen: off + bsize + 1 - 8]
        crc, isize = struct.unpack_from("<II", raw, off + bsize + 1 - 8)
        data = zlib.decompress(cdata, -15)
        if len(data) != isize or (zlib.crc32(data) & 0xFFFFFFFF) != crc:
            raise ValueError("BGZF member failed CRC/ISIZE check")
        out.append(data)
        off += bsize + 1
    return b"".join(out)


def bgzf_compress(data: bytes, block: int = 0xff00, level: int = 6, sizes: list | None = None) -> bytes:
    def member(chunk: bytes) -> bytes:
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        c = co.compress(chunk) + co.flush()
        bsize = len(c) + 25
        return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00"
                + struct.pack("<H", bsize) + c
                + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))
    out = [member(data[i:i + block]) for i in range(0, len(data), block)]
    if sizes is not None:
        sizes.extend(len(m) for m in out)                # compressed size of every data member
    out.append(member(b""))  # EOF marker
    return b"".join(out)


def read_bam(path: str):
    """Returns (header_text, contigs[(name,len)], {tid: Reads}, n_records_total).

    Records with refID == -1 (unplaced) are counted but not returned."""
    d = bgzf_decompress(open(path, "rb").read())
    if d[:4] != b"BAM\x01":
        raise ValueError("bad BAM magic")
    l_text, = struct.unpack_from("<i", d, 4)
    text = d[8:8 + l_text].decode("latin-1")
    p = 8 + l_text
    n_ref, = struct.unpack_from("<i", d, p)
    p += 4
    contigs = []
    for _ in range(n_ref):
        l_name, = struct.unpack_from("<i", d, p)
        name = d[p + 4:p + 4 + l_name - 1].decode()
        l_ref, = struct.unpack_from("<i", d, p + 4 + l_name)
        contigs.append((name, l_ref))
        p += 8 + l_name
    per = {}
    total = 0
    while p < len(d):
        block_size, = struct.unpack_from("<i", d, p)
        (ref_id, pos, l_read_name, mapq, _bin, n_cigar, flag, l_seq, _nref, _npos,
         _tlen) = struct.unpack_from("<iiBBHHHiiii", d, p + 4)
        q = p + 36 + l_read_name
        cig = np.frombuffer(d, dtype="<u4", count=n_cigar, offset=q).copy()
        # long-CIGAR convention (SAMv1 4.2.2): real CIGAR in tag CG:B,I
        if n_cigar == 2 and (cig[0] & 0xF) == 4 and (cig[0] >> 4) == l_seq and (cig[1] & 0xF) == 3:
            t = q + 4 * n_cigar + (l_seq + 1) // 2 + l_seq
            end = p + 4 + block_size
            real = _find_cg(d, t, end)
            if real is not None:
                cig = real
        total += 1
        if ref_id >= 0:
            a = per.setdefault(ref_id, ([], [], [], []))
            a[0].append(pos)
            a[1].append(flag)
            a[2].append(mapq)
            a[3].append(cig)
        p += 4 + block_size
    reads = {}
    for tid, (ps, fl, mq, cg) in per.items():
        off = np.zeros(len(ps) + 1, np.uint32)
        off[1:] = np.cumsum([len(c) for c in cg])
        reads[tid] = Reads(np.asarray(ps, np.int32), np.asarray(fl, np.uint16),
                           np.asarray(mq, np.uint8), off,
                           np.concatenate(cg) if cg else np.zeros(0, np.uint32))
    return text, contigs, reads, total


_TAG_SIZE = {b"A": 1, b"c": 1, b"C": 1, b"s": 2, b"S": 2, b"i": 4, b"I": 4, b"f": 4}


def _find_cg(d: bytes, t: int, end: int):
    while t + 3 <= end:
        tag, typ = d[t:t + 2], d[t + 2:t + 3]
        t += 3
        if typ in _TAG_SIZE:
            t += _TAG_SIZE[typ]
        elif typ in (b"Z", b"H"):
            t = d.index(b"\x00", t) + 1
        elif typ == b"B":
            sub = d[t:t + 1]
            cnt, = struct.unpack_from("<i", d, t + 1)
            t += 5
            if tag == b"CG" and sub == b"I":
                return np.frombuffer(d, dtype="<u4", count=cnt, offset=t).copy()
            t += cnt * _TAG_SIZE[sub]
        else:
            return None
    return None


def _reg2bin(beg: int, end: int) -> int:
    end -= 1
    if beg >> 14 == end >> 14:
        return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17:
        return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20:
        return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23:
        return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26:
        return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def write_bam(path: str, contigs, reads_by_tid, header_text: str | None = None,
              long_cigar_as_cg: bool = True, unplaced: int = 0, level: int = 1, index: bool = False):
    """Write a coordinate-sorted BAM holding the given record streams.

    Sequence/quality are written as l_seq=0 ('*'), which is legal BAM and is
    never consulted by depth.  CIGARs with more than 65535 ops are stored via
    the CG:B,I tag convention when long_cigar_as_cg is set."""
    if header_text is None:
        header_text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join(
            "@SQ\tSN:%s\tLN:%d\n" % (n, l) for n, l in contigs)
    ht = header_text.encode()
    out = [b"BAM\x01", struct.pack("<i", len(ht)), ht, struct.pack("<i", len(contigs))]
    for name, length in contigs:
        nb = name.encode() + b"\x00"
        out.append(struct.pack("<i", len(nb)) + nb + struct.pack("<i", length))
    rid = 0
    cur = sum(len(x) for x in out)                       # uncompressed offset of the next record
    recs = []                                            # (tid, pos, end, offset, offset after) for the index
    for tid in sorted(reads_by_tid):
        r = reads_by_tid[tid]
        for i in range(r.n):
            cig = r.cigar[int(r.cigar_off[i]):int(r.cigar_off[i + 1])]
            ref_len = int(sum(int(c >> 4) for c in cig if (int(c) & 0xF) in (0, 2, 3, 7, 8)))
            name = ("r%d" % rid).encode() + b"\x00"
            rid += 1
            tags = b""
            stored = cig
            if len(cig) > 65535 and long_cigar_as_cg:
                tags = b"CGBI" + struct.pack("<i", len(cig)) + cig.astype("<u4").tobytes()
                stored = np.asarray([(0 << 4) | 4, (ref_len << 4) | 3], np.uint32)  # 0S <ref>N
            pos = int(r.pos[i])
            body = struct.pack("<iiBBHHHiiii", tid, pos, len(name), int(r.mapq[i]),
                               _reg2bin(pos, pos + max(ref_len, 1)), len(stored),
                               int(r.flag[i]), 0, -1, -1, 0)
            body += name + stored.astype("<u4").tobytes() + tags
            out.append(struct.pack("<i", len(body)) + body)
            recs.append((tid, pos, pos + max(ref_len, 1), cur, cur + 4 + len(body)))
            cur += 4 + len(body)
    for _ in range(unplaced):
        name = ("u%d" % rid).encode() + b"\x00"
        rid += 1
        body = struct.pack("<iiBBHHHiiii", -1, -1, len(name), 0, 4680, 0, 4, 0, -1, -1, 0) + name
        out.append(struct.pack("<i", len(body)) + body)
    sizes: list = []
    with open(path, "wb") as fh:
        fh.write(bgzf_compress(b"".join(out), level=level, sizes=sizes))
    if index:
        write_bai(path + ".bai", len(contigs), recs, sizes)


def write_bai(path: str, n_ref: int, recs, member_sizes, block: int = 0xff00):
    """A .bai (SAMv1 5.2) for records (tid, pos, end, uncompressed offset, offset after) of a
    file whose data members all hold `block` uncompressed bytes: per reference the binning
    index (one merged chunk per bin) and the 16 kb linear index (virtual offset of the first
    record overlapping each window; empty leading windows 0, later ones inherit)."""
    coff = np.concatenate([[0], np.cumsum(member_sizes)]).astype(np.int64)
    voff = lambda o: (int(coff[o // block]) << 16) | (o % block)
    bins = [dict() for _ in range(n_ref)]
    lin = [dict() for _ in range(n_ref)]
    for tid, pos, end, o0, o1 in recs:
        b = _reg2bin(pos, end)
        v0, v1 = voff(o0), voff(o1)
        lo, hi = bins[tid].get(b, (v0, v1))
        bins[tid][b] = (min(lo, v0), max(hi, v1))
        for w in range(pos >> 14, ((end - 1) >> 14) + 1):
            lin[tid].setdefault(w, v0)
    out = [b"BAI\x01", struct.pack("<i", n_ref)]
    for tid in range(n_ref):
        out.append(struct.pack("<i", len(bins[tid])))
        for b in sorted(bins[tid]):
            out.append(struct.pack("<Ii", b, 1) + struct.pack("<QQ", *bins[tid][b]))
        n_intv = max(lin[tid]) + 1 if lin[tid] else 0
        out.append(struct.pack("<i", n_intv))
        last = 0
        for w in range(n_intv):
            last = lin[tid].get(w, last)
            out.append(struct.pack("<Q", last))
    with open(path, "wb") as fh:
        fh.write(b"".join(out))


def read_bai_linear(path: str):
    """{tid: sorted unique non-zero linear-index virtual offsets} of a .bai."""
    d = open(path, "rb").read()
    assert d[:4] == b"BAI\x01"
    n_ref, = struct.unpack_from("<i", d, 4)
    p = 8
    res = {}
    for tid in range(n_ref):
        n_bin, = struct.unpack_from("<i", d, p)
        p += 4
        for _ in range(n_bin):
            _b, n_chunk = struct.unpack_from("<Ii", d, p)
            p += 8 + 16 * n_chunk
        n_intv, = struct.unpack_from("<i", d, p)
        p += 4
        io = np.frombuffer(d, "<u8", n_intv, p)
        p += 8 * n_intv
        res[tid] = np.unique(io[io != 0])
    return res
