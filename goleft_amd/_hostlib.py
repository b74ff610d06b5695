"""ctypes binding of libgoleft_host.so (include/goleft_depth_host.h): the C++
host side of `goleft depth` (BAM decode, tiling, BED rows, CLI entry)."""
from __future__ import annotations

import ctypes as C
import os

from . import _lib
from ._lib import GdRun

HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(HERE, "libgoleft_host.so")
_P = C.c_void_p

SYMBOLS = {
    "gdh_depth_main": (C.c_int, [C.c_int, C.POINTER(C.c_char_p)]),
    "gdh_chrom_start_end": (C.c_int, [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t,
                                      C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "gdh_step": (C.c_int64, [C.c_int32]),
    "gdh_lpt_assign": (C.c_int, [_P, C.c_size_t, _P, C.c_size_t, C.c_size_t, _P]),
    "gdh_format_region": (C.c_int, [C.c_char_p, C.c_int64, C.c_int64, C.c_int32, _P, C.c_size_t,
                                    _P, C.c_size_t, C.c_char_p, C.c_char_p]),
    "gdh_depthwed_main": (C.c_int, [C.c_int, C.POINTER(C.c_char_p)]),
    "gdh_depthwed_run": (C.c_int, [C.c_int64, C.POINTER(C.c_char_p), C.c_int, C.c_char_p]),
    "gdh_depthwed_cells": (None, [_P, _P, C.c_size_t, _P]),
    "gdh_multidepth_main": (C.c_int, [C.c_int, C.POINTER(C.c_char_p)]),
    "gdh_multidepth_run": (C.c_int, [C.c_int, C.POINTER(C.c_char_p), C.c_char_p]),
    "gdh_plan_ingest_passes": (C.c_size_t, [_P, _P, C.c_size_t, _P, C.c_size_t, C.c_uint64, C.c_uint64, C.c_size_t,
                                            _P, _P, _P, _P]),
    "gdh_plan_ingest_parts": (C.c_size_t, [_P, C.c_size_t, C.c_uint64, C.c_uint64, C.c_size_t, _P, _P, _P, _P, _P]),
    "gdh_multidepth_blocks": (C.c_int64, [_P, _P, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                          _P, _P, C.c_int64]),
    "gdh_bam_open": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(_P)]),
    "gdh_bam_close": (None, [_P]),
    "gdh_bam_error": (C.c_char_p, [_P]),
    "gdh_bam_n_contigs": (C.c_int, [_P]),
    "gdh_bam_contig_name": (C.c_char_p, [_P, C.c_int]),
    "gdh_bam_contig_length": (C.c_int64, [_P, C.c_int]),
    "gdh_bam_seek_contig": (C.c_int, [_P, C.c_int]),
    "gdh_bam_next": (C.c_int, [_P, C.c_size_t, C.POINTER(C.c_int32), C.POINTER(C.c_size_t),
                               C.POINTER(C.c_size_t), C.POINTER(_P), C.POINTER(_P), C.POINTER(_P),
                               C.POINTER(_P), C.POINTER(_P)]),
    "gdh_bam_n_records": (C.c_uint64, [_P]),
    "gdh_intervals_read": (C.c_int, [C.POINTER(C.c_char_p), C.c_int, C.POINTER(_P)]),
    "gdh_intervals_free": (None, [_P]),
    "gdh_intervals_overlaps": (C.c_int, [_P, C.c_char_p, C.c_int64, C.c_int64]),
    "gdh_intervals_count": (C.c_size_t, [_P, C.c_char_p]),
    "gdh_list_members": (C.c_int64, [_P, C.c_size_t, C.c_uint64, _P, C.c_size_t, C.c_uint, C.c_size_t, C.c_size_t,
                                     _P, _P, _P, _P, _P]),
    "gdh_list_members_fd": (C.c_int64, [C.c_int, C.c_uint64, C.c_size_t, _P, C.c_size_t, C.c_uint, C.c_size_t, C.c_size_t,
                                        _P, _P, _P, _P, _P]),
    "gdh_samtools_main": (C.c_int, [C.c_int, C.POINTER(C.c_char_p)]),
    "gdh_produce_in_place": (C.c_int, [_P, C.c_int32, _P, _P, _P, _P, _P, C.c_size_t, C.c_size_t, C.c_int, C.c_size_t]),
    "gdh_set_fast_exit": (C.c_int, [C.c_int]),
    "gdh_get_fast_exit": (C.c_int, []),
    "gdh_set_stats_contract": (C.c_int, [C.c_int]),
    "gdh_get_stats_contract": (C.c_int, []),
    "gdh_format_stats": (C.c_int, [C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_uint32, C.c_uint32, C.c_uint32,
                                   C.c_uint32, C.c_uint32, C.c_char_p, C.c_size_t]),
}

# the forks of the `--stats` contract (include/goleft_depth_host.h)
STATS_DENOM_ACGT, STATS_MASKED_ACGT, STATS_CPG_CLAMP, STATS_CPG_RAW_LINES = 1, 2, 4, 8
STATS_WINDOW, STATS_FAIDX = 0, 15

_LIB = None


def load():
    global _LIB
    if _LIB is not None:
        return _LIB
    _lib.load()   # the host library links against the device library
    if not os.path.exists(SO_PATH):
        raise ImportError("goleft_amd: %s is missing -- run __graft_entry__.build()" % SO_PATH)
    lib = C.CDLL(SO_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


def set_stats_contract(contract: int) -> None:
    """gdh_set_stats_contract: which reading of faidx.Stats `goleft depth --stats` prints (STATS_* above)."""
    if load().gdh_set_stats_contract(int(contract)) != 0:
        raise ValueError("stats contract %r" % (contract,))


def get_stats_contract() -> int:
    return int(load().gdh_get_stats_contract())


def format_stats(contract, known, start, end, n_gc, n_cpg, n_masked, n_acgt, n_masked_acgt) -> str:
    buf = C.create_string_buffer(128)
    load().gdh_format_stats(int(contract), int(known), int(start), int(end), int(n_gc), int(n_cpg), int(n_masked),
                            int(n_acgt), int(n_masked_acgt), buf, 128)
    return buf.value.decode()


def chrom_start_end(line: bytes):
    """depth/depth.go:75-94."""
    chrom = C.create_string_buffer(4096)
    s, e = C.c_int64(), C.c_int64()
    if load().gdh_chrom_start_end(line, len(line), chrom, 4096, C.byref(s), C.byref(e)) != 0:
        raise ValueError("couldn't get region from line %r" % line)
    return chrom.value.decode(), s.value, e.value


def lpt_assign(tids, lengths, n_shards):
    """gdh_lpt_assign: the shard (engine context / device) of every tid, as `goleft depth` assigns them."""
    import numpy as np
    t = np.ascontiguousarray(tids, np.int32)
    l = np.ascontiguousarray(lengths, np.int64)
    out = np.full(len(t), -1, np.int32)
    rc = load().gdh_lpt_assign(t.ctypes.data, len(t), l.ctypes.data, len(l), int(n_shards), out.ctypes.data)
    if rc != 0:
        raise ValueError("gdh_lpt_assign: bad arguments")
    return out


def format_region(chrom, start, end, W, sums, runs, depth_path, callable_path):
    import numpy as np
    sums = np.ascontiguousarray(sums, np.int64)
    runs = np.ascontiguousarray(runs, np.int32).reshape(-1, 3)
    rc = load().gdh_format_region(chrom.encode(), start, end, W, sums.ctypes.data, len(sums),
                                  runs.ctypes.data, len(runs), depth_path.encode(),
                                  callable_path.encode())
    if rc != 0:
        raise OSError("gdh_format_region failed")


def read_bam(path: str, threads: int = 0, max_reads: int = 1 << 20, seek_tid=None):
    """Decode a BAM with the C++ reader -> (contigs, {tid: (pos, flag, mapq, off, cigar)}, n_records)."""
    import numpy as np
    lib = load()
    h = _P()
    rc = lib.gdh_bam_open(path.encode(), threads, C.byref(h))
    try:
        if rc != 0:
            raise OSError(lib.gdh_bam_error(h).decode() if h else "open failed")
        contigs = [(lib.gdh_bam_contig_name(h, i).decode(), lib.gdh_bam_contig_length(h, i))
                   for i in range(lib.gdh_bam_n_contigs(h))]
        if seek_tid is not None:
            lib.gdh_bam_seek_contig(h, seek_tid)
        parts = {}
        tid, n, m = C.c_int32(), C.c_size_t(), C.c_size_t()
        pp, pf, pm, po, pc = _P(), _P(), _P(), _P(), _P()
        while True:
            rc = lib.gdh_bam_next(h, max_reads, C.byref(tid), C.byref(n), C.byref(m), C.byref(pp),
                                  C.byref(pf), C.byref(pm), C.byref(po), C.byref(pc))
            if rc < 0:
                raise OSError(lib.gdh_bam_error(h).decode())
            if rc == 0:
                break

            def arr(ptr, dt, cnt):
                if cnt == 0:
                    return np.zeros(0, dt)
                return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)),
                                             (cnt * np.dtype(dt).itemsize,)).view(dt).copy()
            parts.setdefault(tid.value, []).append(
                (arr(pp, np.int32, n.value), arr(pf, np.uint16, n.value), arr(pm, np.uint8, n.value),
                 arr(po, np.uint32, n.value + 1), arr(pc, np.uint32, m.value)))
        out = {}
        for t, ps in parts.items():
            pos = np.concatenate([p[0] for p in ps])
            flag = np.concatenate([p[1] for p in ps])
            mapq = np.concatenate([p[2] for p in ps])
            cig = np.concatenate([p[4] for p in ps])
            offs, base = [np.zeros(1, np.uint32)], 0
            for p in ps:
                offs.append(p[3][1:] + np.uint32(base))
                base += len(p[4])
            out[t] = (pos, flag, mapq, np.concatenate(offs), cig)
        return contigs, out, int(lib.gdh_bam_n_records(h))
    finally:
        if h:
            lib.gdh_bam_close(h)


class Intervals:
    """depth/intervals.go ReadTree / Overlaps."""

    def __init__(self, *paths: str):
        self._lib = load()
        self._h = _P()
        arr = (C.c_char_p * len(paths))(*[p.encode() for p in paths])
        if self._lib.gdh_intervals_read(arr, len(paths), C.byref(self._h)) != 0:
            raise OSError("ReadTree failed")

    def overlaps(self, chrom: str, start: int, end: int) -> bool:
        return bool(self._lib.gdh_intervals_overlaps(self._h, chrom.encode(), start, end))

    def count(self, chrom: str) -> int:
        return int(self._lib.gdh_intervals_count(self._h, chrom.encode()))

    def __del__(self):
        try:
            if self._h:
                self._lib.gdh_intervals_free(self._h)
        except Exception:
            pass


def multidepth_blocks(any_mask, suf_mask, chunk, max_skip=10, min_size=15, window=10000000):
    """The multidepth block state machine (multidepth/multidepth.go:188-268) over boolean
    site masks, found on the device (gd_md_load_flags + gd_md_blocks); returns an int64 array
    [n_blocks, 2] of {start, end}."""
    import numpy as np
    a = np.asarray(any_mask, bool)
    s = np.asarray(suf_mask, bool)
    n = a.size
    pack = lambda m: np.packbits(np.concatenate([m, np.zeros((-n) % 32, bool)]), bitorder="little").view(np.uint32)
    aw, sw = pack(a), pack(s)
    cnt = load().gdh_multidepth_blocks(aw.ctypes.data, sw.ctypes.data, n, chunk, max_skip, min_size, window,
                                       None, None, 0)
    if cnt < 0:
        raise ValueError("gdh_multidepth_blocks: bad arguments" if cnt == -1 else "gdh_multidepth_blocks: no usable device")
    st = np.zeros(cnt, np.int64)
    en = np.zeros(cnt, np.int64)
    load().gdh_multidepth_blocks(aw.ctypes.data, sw.ctypes.data, n, chunk, max_skip, min_size, window,
                                 st.ctypes.data, en.ctypes.data, cnt)
    return np.stack([st, en], 1)


def list_members(data: bytes, beg: int, member_starts, threads: int = 16, min_bytes: int = 64 << 20):
    """gdh_list_members: (off u64[n], size u32[n], hdr u16[n], isize u32[n], crc u32[n]) of the complete BGZF members
    of `data` (file offset `beg`), the walk cut at .bai-known member starts and run by `threads` threads."""
    import numpy as np
    raw = np.frombuffer(data, np.uint8)
    st = np.ascontiguousarray(member_starts, np.uint64)
    args = (raw.ctypes.data, raw.size, int(beg), st.ctypes.data if st.size else None, st.size, int(threads), int(min_bytes))
    n = load().gdh_list_members(*args, 0, None, None, None, None, None)
    if n < 0:
        raise ValueError("not a BGZF range")
    off = np.zeros(n, np.uint64); size = np.zeros(n, np.uint32); hdr = np.zeros(n, np.uint16)
    isize = np.zeros(n, np.uint32); crc = np.zeros(n, np.uint32)
    load().gdh_list_members(*args, n, off.ctypes.data, size.ctypes.data, hdr.ctypes.data, isize.ctypes.data, crc.ctypes.data)
    return off, size, hdr, isize, crc


def list_members_fd(path: str, beg: int, n_bytes: int, member_starts, threads: int = 16, min_bytes: int = 64 << 20):
    """gdh_list_members_fd: the same table, the n_bytes from offset `beg` of the file read with pread."""
    import os
    import numpy as np
    st = np.ascontiguousarray(member_starts, np.uint64)
    fd = os.open(path, os.O_RDONLY)
    try:
        args = (fd, int(beg), int(n_bytes), st.ctypes.data if st.size else None, st.size, int(threads), int(min_bytes))
        n = load().gdh_list_members_fd(*args, 0, None, None, None, None, None)
        if n < 0:
            raise ValueError("not a BGZF range")
        off = np.zeros(n, np.uint64); size = np.zeros(n, np.uint32); hdr = np.zeros(n, np.uint16)
        isize = np.zeros(n, np.uint32); crc = np.zeros(n, np.uint32)
        load().gdh_list_members_fd(*args, n, off.ctypes.data, size.ctypes.data, hdr.ctypes.data, isize.ctypes.data, crc.ctypes.data)
        return off, size, hdr, isize, crc
    finally:
        os.close(fd)


def plan_ingest_passes(start, has, wanted, file_size, group_bytes):
    """[(first, last, beg, end)] -- how goleft-depth cuts a BAM into device passes (gdh_plan_ingest_passes)."""
    import numpy as np
    st = np.ascontiguousarray(start, np.uint64)
    hs = np.ascontiguousarray(has, np.uint8)
    w = np.ascontiguousarray(wanted, np.int32)
    cap = max(1, len(w))
    f, l, b, e = (np.zeros(cap, np.uint64) for _ in range(4))
    n = load().gdh_plan_ingest_passes(st.ctypes.data, hs.ctypes.data, len(st), w.ctypes.data, len(w), file_size,
                                      group_bytes, cap, f.ctypes.data, l.ctypes.data, b.ctypes.data, e.ctypes.data)
    return [(int(f[k]), int(l[k]), int(b[k]), int(e[k])) for k in range(n)]


def plan_ingest_parts(anchors, end, part_bytes):
    """[(a_lo, a_hi, beg, end, scale)] -- how one reference is read in parts cut at .bai anchors (gdh_plan_ingest_parts)."""
    import numpy as np
    an = np.ascontiguousarray(anchors, np.uint64)
    cap = len(an) + 1
    lo, hi, b, e = (np.zeros(cap, np.uint64) for _ in range(4))
    sc = np.zeros(cap, np.float64)
    n = load().gdh_plan_ingest_parts(an.ctypes.data, len(an), end, part_bytes, cap, lo.ctypes.data, hi.ctypes.data,
                                     b.ctypes.data, e.ctypes.data, sc.ctypes.data)
    return [(int(lo[k]), int(hi[k]), int(b[k]), int(e[k]), float(sc[k])) for k in range(n)]
