"""Python twin of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Parity status: "parity unpinned" -- see oracle/depth_oracle.h.  This module
holds (a) an independent brute-force per-position counter used to cross-check
the C restatement, (b) a second, independent restatement of the reference's
`callback` reducer (/root/reference/depth/depth.go:238-364) that emits the BED
rows as strings, and (c) a ctypes binding of oracle/libdepth_oracle.so.

Nothing under goleft_amd/ may import this module.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from dataclasses import dataclass

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_FLAG_MASK = 0x704  # UNMAP|SECONDARY|QCFAIL|DUP (samtools depth default)

# BAM cigar op codes: MIDNSHP=X
OP_M, OP_I, OP_D, OP_N, OP_S, OP_H, OP_P, OP_EQ, OP_X = range(9)
_CONSUMES_REF = {OP_M, OP_D, OP_N, OP_EQ, OP_X}
_COUNTED = {OP_M, OP_EQ, OP_X}
CLASS_NAMES = ("NO_COVERAGE", "LOW_COVERAGE", "CALLABLE", "EXCESSIVE_COVERAGE")


@dataclass
class Reads:
    """One contig's decoded records (coordinate sorted), SoA."""
    pos: np.ndarray        # int32 [n]
    flag: np.ndarray       # uint16 [n]
    mapq: np.ndarray       # uint8 [n]
    cigar_off: np.ndarray  # uint32 [n+1]
    cigar: np.ndarray      # uint32 [m]

    def __post_init__(self):
        self.pos = np.ascontiguousarray(self.pos, dtype=np.int32)
        self.flag = np.ascontiguousarray(self.flag, dtype=np.uint16)
        self.mapq = np.ascontiguousarray(self.mapq, dtype=np.uint8)
        self.cigar_off = np.ascontiguousarray(self.cigar_off, dtype=np.uint32)
        self.cigar = np.ascontiguousarray(self.cigar, dtype=np.uint32)
        assert self.cigar_off.shape[0] == self.pos.shape[0] + 1

    @property
    def n(self) -> int:
        return int(self.pos.shape[0])

    @property
    def n_ops(self) -> int:
        return int(self.cigar.shape[0])

    def slice(self, lo: int, hi: int) -> "Reads":
        off = self.cigar_off[lo:hi + 1].astype(np.int64)
        return Reads(self.pos[lo:hi], self.flag[lo:hi], self.mapq[lo:hi],
                     (off - off[0]).astype(np.uint32),
                     self.cigar[off[0]:off[-1]])


# --------------------------------------------------------------------------
# (a) independent brute force: a dict-free, loop-per-base counter
# --------------------------------------------------------------------------
def perbase_bruteforce(r: Reads, q: int, start: int, end: int,
                       flag_mask: int = DEFAULT_FLAG_MASK) -> np.ndarray:
    """samtools>=1.13 `depth -Q q` semantics, one Python loop per base."""
    out = [0] * max(0, end - start)
    for i in range(r.n):
        if int(r.flag[i]) & flag_mask:
            continue
        if int(r.mapq[i]) < q:
            continue
        cur = int(r.pos[i])
        for k in range(int(r.cigar_off[i]), int(r.cigar_off[i + 1])):
            c = int(r.cigar[k])
            op, ln = c & 0xF, c >> 4
            if op in _COUNTED:
                for p in range(max(cur, start), min(cur + ln, end)):
                    out[p - start] += 1
            if op in _CONSUMES_REF:
                cur += ln
    return np.asarray(out, dtype=np.int32)


def perbase_numpy(r: Reads, q: int, start: int, end: int,
                  flag_mask: int = DEFAULT_FLAG_MASK) -> np.ndarray:
    """Vectorised twin (diff marks + cumsum) for mid-sized inputs."""
    L = max(0, end - start)
    if L == 0:
        return np.zeros(0, np.int32)
    nops = np.diff(r.cigar_off.astype(np.int64))
    read_of_op = np.repeat(np.arange(r.n), nops)
    op = (r.cigar & 0xF).astype(np.int64)
    ln = (r.cigar >> 4).astype(np.int64)
    consumes = np.isin(op, list(_CONSUMES_REF))
    adv = np.where(consumes, ln, 0)
    cum = np.cumsum(adv) - adv                      # exclusive over all ops
    first = r.cigar_off[:-1].astype(np.int64)
    safe_first = np.minimum(first, max(len(cum) - 1, 0))
    base = np.where(nops > 0, cum[safe_first] if len(cum) else 0, 0)
    ref_start = r.pos.astype(np.int64)[read_of_op] + cum - base[read_of_op]
    keep = ((r.flag.astype(np.int64) & flag_mask) == 0) & (r.mapq.astype(np.int64) >= q)
    counted = np.isin(op, list(_COUNTED)) & keep[read_of_op] & (ln > 0)
    s = np.clip(ref_start[counted], start, end)
    e = np.clip(ref_start[counted] + ln[counted], start, end)
    ok = s < e
    diff = np.zeros(L + 1, np.int64)
    np.add.at(diff, s[ok] - start, 1)
    np.add.at(diff, e[ok] - start, -1)
    return np.cumsum(diff[:L]).astype(np.int32)


# --------------------------------------------------------------------------
# (b) second restatement of depth/depth.go:238-364 (string output)
# --------------------------------------------------------------------------
def cov_class(depth: int, mincov: int, maxmean: int) -> str:
    """depth/depth.go:223-234."""
    if depth == 0:
        return "NO_COVERAGE"
    if depth < mincov:
        return "LOW_COVERAGE"
    if maxmean > 0 and depth >= maxmean:
        return "EXCESSIVE_COVERAGE"
    return "CALLABLE"


def fmt_g4(x: float) -> str:
    """Go %.4g == C %.4g."""
    return "%.4g" % x


def callback_py(chrom: str, region_start: int, region_end: int,
                depth: np.ndarray, W: int, mincov: int, maxmean: int):
    """Returns (depth_bed_rows, callable_bed_rows) as lists of str (no \\n).

    `depth[i]` is the depth at region_start+i; only positions with depth>0
    are 'printed by samtools'."""
    hd, ca = [], []
    cache_sum, cache_len = 0.0, 0
    pos = 0
    last_window = max(0, region_start // W)
    c0 = c1 = region_start - 1
    last_cls = ""

    def mean(l):
        if cache_len == 0 or l == 0:
            return 0.0
        return cache_sum / float(l)

    covered = np.nonzero(np.asarray(depth) > 0)[0]
    for idx in covered:
        pos = region_start + int(idx)
        d = int(depth[idx])
        if pos // W != last_window:
            this_window = pos // W
            for iw in range(last_window, this_window):
                s = max(region_start, iw * W)
                e = min(region_end, (iw + 1) * W)
                hd.append("%s\t%d\t%d\t%s" % (chrom, s, e, fmt_g4(mean(e - s))))
                cache_sum, cache_len = 0.0, 0
            last_window = this_window
        cache_sum += float(d)
        cache_len += 1
        cls = cov_class(d, mincov, maxmean)
        if cls != last_cls or pos != c1 + 1:
            if last_cls != "":
                ca.append("%s\t%d\t%d\t%s" % (chrom, c0, c1 + 1, last_cls))
            if pos != c1 + 1:
                ca.append("%s\t%d\t%d\t%s" % (chrom, c1 + 1, pos, "NO_COVERAGE"))
            last_cls = cls
            c0 = c1 = pos
        else:
            c1 = pos
    if c0 != -1 and last_cls != "":
        ca.append("%s\t%d\t%d\t%s" % (chrom, c0, c1 + 1, last_cls))
    if cache_len > 0:
        s = pos // W * W
        if s < region_end:
            s = max(s, region_start)
            e = min(region_end, s + W)
            hd.append("%s\t%d\t%d\t%s" % (chrom, s, e, fmt_g4(mean(e - s))))
            cache_sum, cache_len = 0.0, 0
            pos = e
    if c1 + 1 < region_end:
        if c1 != -1:
            ca.append("%s\t%d\t%d\tNO_COVERAGE" % (chrom, c1 + 1, region_end))
        else:
            ca.append("%s\t%d\t%d\tNO_COVERAGE" % (chrom, region_start, region_end))
        ds = max(region_start, pos) // W * W
        while ds < region_end and pos < region_end:
            de = min(region_end, ds + W)
            s = max(ds, region_start)
            hd.append("%s\t%d\t%d\t%s" % (chrom, s, de, fmt_g4(mean(de - s))))
            cache_sum, cache_len = 0.0, 0
            ds += W
    return hd, ca


def seq_stats(seq: bytes, start: int, end: int):
    """(n_gc, n_cpg, n_masked) of seq[start:end] clipped to the contig -- the integer
    counts behind the `--stats` columns (depth/depth.go:191-200 -> faidx.Stats, an
    external module at go.mod:12 whose values no reference test asserts: PARITY
    UNPINNED; this restatement IS the contract of gd_seq_stats and of
    host/fasta_stats.hpp).  Pure Python, byte by byte."""
    s, e = max(0, start), min(end, len(seq))
    gc = cpg = low = 0
    for i in range(s, e):
        c = seq[i]
        if c in b"GCgc":
            gc += 1
        if 0x61 <= c <= 0x7a:
            low += 1
        if c in b"Cc" and i + 1 < len(seq) and seq[i + 1] in b"Gg":
            cpg += 1
    return gc, cpg, low


# the forks of the `--stats` contract (include/goleft_depth_host.h GDH_STATS_*)
STATS_DENOM_ACGT, STATS_MASKED_ACGT, STATS_CPG_CLAMP, STATS_CPG_RAW_LINES = 1, 2, 4, 8
STATS_WINDOW, STATS_FAIDX = 0, 15


def seq_counts(seq: bytes, start: int, end: int, line_bases: int = 0):
    """(n_gc, n_cpg, n_masked, n_acgt, n_masked_acgt) of seq[start:end] clipped to the contig -- seq_stats plus
    what the other reading of faidx.Stats needs: A/C/G/T bases of either case, lower-case a/c/g/t; with
    line_bases > 0 (a scan of the window's raw, line-broken bytes) a C that is the last base of a FASTA line
    (position % line_bases == line_bases - 1) or the last base of the WINDOW starts no CpG: neither G is among the
    bytes scanned.  Pure Python, byte by byte; the contract of gd_seq_stats_ex."""
    s, e = max(0, start), min(end, len(seq))
    gc = cpg = low = acgt = lacgt = 0
    for i in range(s, e):
        c = seq[i]
        if c in b"GCgc":
            gc += 1
        if 0x61 <= c <= 0x7a:
            low += 1
        if c in b"ACGTacgt":
            acgt += 1
            if c in b"acgt":
                lacgt += 1
        eol = line_bases > 0 and (i % line_bases == line_bases - 1 or i == e - 1)
        if c in b"Cc" and not eol and i + 1 < len(seq) and seq[i + 1] in b"Gg":
            cpg += 1
    return gc, cpg, low, acgt, lacgt


def fmt_g3(x: float) -> str:
    """Go's %.3g for the magnitudes --stats prints (fractions in [0, 2])."""
    t = "%.3g" % x
    if "e" in t:                                  # Go prints at least two exponent digits too
        m, ex = t.split("e")
        t = "%se%s%02d" % (m, ex[0], int(ex[1:]))
    return t


def stats_columns(seq: bytes, start: int, end: int, contract: int = STATS_FAIDX, line_bases: int = 0) -> str:
    """"\tGC\tCpG\tMasked" as getStats formats them (depth/depth.go:199) under one of the contracts of
    include/goleft_depth_host.h (gdh_format_stats is the product's twin of this function)."""
    gc, cpg, low, acgt, lacgt = seq_counts(seq, start, end, line_bases if contract & STATS_CPG_RAW_LINES else 0)
    tot = float(acgt) if contract & STATS_DENOM_ACGT else float(end - start)
    if start >= len(seq) or end <= start or tot == 0:
        return "\t0\t0\t0"
    c = 2.0 * cpg / tot
    if contract & STATS_CPG_CLAMP:
        c = min(1.0, c)
    m = (lacgt if contract & STATS_MASKED_ACGT else low) / tot
    return "\t%s\t%s\t%s" % (fmt_g3(gc / tot), fmt_g3(c), fmt_g3(m))


# ---------------------------------------------------------------------------
# multidepth (multidepth/multidepth.go), restated line by line.  The per-position
# stream is what `samtools depth -q 0 -Q Q -d MaxCov -r chrom:start bams...`
# (:203-207) prints: one row per position >= start at which at least one BAM has
# depth > 0 (samtools >= 1.13 semantics; PARITY UNPINNED).
# ---------------------------------------------------------------------------
def md_means(sites, depth_rows):
    """means (:270-283): sites = positions, depth_rows[k] = the S depths at sites[k]."""
    dps = [0.0] * len(depth_rows[0])
    for row in depth_rows:
        for i, d in enumerate(row):
            dps[i] += float(int(d)) / 1000.
    l = sites[-1] - sites[0] + 1
    return ["%.2f" % (d / float(l) * 1000) for d in dps]


def md_split_blocks(chrom, cache, depth_of, window):
    """splitBlocks (:188-201); cache = sufficient positions, depth_of(pos) -> S depths."""
    blocks = []
    i = lasti = 0
    while i < len(cache):
        start = cache[i]
        i += 1
        while i < len(cache) and cache[i] - start < window:
            i += 1
        end = cache[i - 1] + 1
        sub = cache[lasti:i]
        blocks.append("%s\t%d\t%d\t%s" % (chrom, start, end, "\t".join(md_means(sub, [depth_of(p) for p in sub]))))
        lasti = i
    return blocks


def multidepth_py(chrom, depths, mincov=7, maxskip=10, minsize=15, window=10000000, min_samples=0.5,
                  chunk_size=5000000):
    """Block lines (without the header) in chunk order, i.e. what `multidepth -p 1` prints.
    depths: S equally long int arrays (the per-base depth of each BAM at -Q)."""
    D = np.stack([np.asarray(d, np.int64) for d in depths])
    S, L = D.shape
    if S > 50:
        chunk_size //= 5                                   # :62-64
    need = int(0.5 + min_samples * float(S))               # :66
    printed = np.flatnonzero((D > 0).any(0))
    suf = (D >= mincov).sum(0) > need                      # sufficientDepth :163-171 (strictly more)
    depth_of = lambda p: D[:, p]
    out = []
    for i in range(0, L, chunk_size):                      # genRegions :130-141
        rstart = i + 1                                     # 1-based
        cache, blocks = [], []
        seen0 = False
        for p in printed[np.searchsorted(printed, i):]:    # aggregate :203-268
            p = int(p)
            s = bool(suf[p])
            if not s:
                seen0 = True
                if p > rstart + chunk_size:
                    if len(cache) == 0 or p - cache[-1] >= maxskip:
                        break
            if not seen0:
                continue
            if (len(cache) == 0 or p - (cache[-1] + 1) <= maxskip) and s:
                cache.append(p)
            elif len(cache) > 0 and p - (cache[-1] + 1) > maxskip:
                if len(cache) >= minsize:
                    blocks += md_split_blocks(chrom, cache, depth_of, window)
                cache = []
                if s:
                    cache.append(p)
        if cache:
            blocks += md_split_blocks(chrom, cache, depth_of, window)
        out += blocks
    return out


def md_short_name(path: str, rg_samples=()) -> str:
    """indexcov.GetShortName(b, false) (indexcov/indexcov.go:213-246): the single @RG SM
    value when there is exactly one, else from the file name."""
    sm = set(rg_samples)
    if len(sm) > 1:
        raise ValueError("bam reagroup: more than one RG for %s" % path)
    if len(sm) == 1:
        return next(iter(sm))
    v = path.split("/")[-1].split(".")
    return v[0] if len(v) <= 2 else "-".join(v[:-1])


def step_for(W: int) -> int:
    """depth/depth.go:48,:132."""
    return max(1, 10000000 // W) * W


def tiles_for(length: int, W: int):
    """depth/depth.go:150-154 as 0-based half-open (start, end)."""
    step = step_for(W)
    return [(i, min(i + step, length)) for i in range(0, length, step)]


# --------------------------------------------------------------------------
# (c) ctypes binding of the C restatement
# --------------------------------------------------------------------------
class _GdoReads(ctypes.Structure):
    _fields_ = [("pos", ctypes.c_void_p), ("flag", ctypes.c_void_p),
                ("mapq", ctypes.c_void_p), ("cigar_off", ctypes.c_void_p),
                ("cigar", ctypes.c_void_p), ("n", ctypes.c_size_t)]


_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(HERE, "libdepth_oracle.so")
    src = os.path.join(HERE, "depth_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", HERE, "-B", "libdepth_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        L.gdo_perbase.argtypes = [ctypes.POINTER(_GdoReads), ctypes.c_int, ctypes.c_uint32,
                                  ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]
        L.gdo_perbase.restype = None
        L.gdo_perbase_diff.argtypes = L.gdo_perbase.argtypes
        L.gdo_perbase_diff.restype = None
        L.gdo_cov_class.argtypes = [ctypes.c_int] * 3
        L.gdo_cov_class.restype = ctypes.c_int
        L.gdo_chrom_start_end.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p,
                                          ctypes.c_size_t, ctypes.POINTER(ctypes.c_long),
                                          ctypes.POINTER(ctypes.c_long)]
        L.gdo_chrom_start_end.restype = ctypes.c_int
        L.gdo_step.argtypes = [ctypes.c_int]
        L.gdo_step.restype = ctypes.c_long
        L.gdo_tiles.argtypes = [ctypes.c_long, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                ctypes.c_size_t]
        L.gdo_tiles.restype = ctypes.c_size_t
        L.gdo_callback_append.argtypes = [ctypes.c_char_p, ctypes.c_long, ctypes.c_long,
                                          ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p]
        L.gdo_callback_append.restype = ctypes.c_int
        _LIB = L
    return _LIB


def _as_struct(r: Reads) -> _GdoReads:
    return _GdoReads(r.pos.ctypes.data, r.flag.ctypes.data, r.mapq.ctypes.data,
                     r.cigar_off.ctypes.data, r.cigar.ctypes.data, r.n)


def perbase_c(r: Reads, q: int, start: int, end: int,
              flag_mask: int = DEFAULT_FLAG_MASK, diff: bool = False) -> np.ndarray:
    out = np.zeros(max(0, end - start), np.int32)
    st = _as_struct(r)
    fn = lib().gdo_perbase_diff if diff else lib().gdo_perbase
    fn(ctypes.byref(st), q, flag_mask, start, end, out.ctypes.data)
    return out


def read_ends(r: Reads) -> np.ndarray:
    """Reference position after the last reference-consuming op of every read (int64)."""
    op = r.cigar & 0xF
    cons = np.where((op == 0) | (op == 2) | (op == 3) | (op == 7) | (op == 8), (r.cigar >> 4).astype(np.int64), 0)
    cs = np.zeros(cons.shape[0] + 1, np.int64)
    np.cumsum(cons, out=cs[1:])
    off = r.cigar_off.astype(np.int64)
    return r.pos.astype(np.int64) + cs[off[1:]] - cs[off[:-1]]


def tiled_contig_check(r: Reads, q: int, length: int, W: int, mincov: int, maxmean: int, step: int,
                       workers: int, got: np.ndarray = None, tile: int = 10_000_000):
    """One whole contig through the C oracle (gdo_perbase_diff) in W-aligned tiles of ~`tile` positions, `workers`
    tiles at a time (ctypes and the numpy reductions release the GIL) -- what makes a 3.1 Gb genome affordable:
    the per-base vector of a tile is compared with got[s:e] right away (never held for the whole contig), and the
    tile's window sums / minima and class-run starts (depth/depth.go:293-323: a break where the class changes and at
    multiples of `step`) are kept.  Reads that start before a tile but reach into it are found through the running
    maximum of the read ends, so any read length is handled.
    -> dict(equal, first_diff, sums int64[nw], mins int32[nw], run_starts int64[k], run_cls int8[k])"""
    from concurrent.futures import ThreadPoolExecutor
    lib()
    tile = max(W, tile // W * W)
    cmax = np.maximum.accumulate(read_ends(r)) if r.n else np.zeros(0, np.int64)
    jobs = [(s, min(length, s + tile)) for s in range(0, length, tile)]

    def one(j):
        s, e = jobs[j]
        lo = int(np.searchsorted(cmax, s, "right"))       # reads before it end at or before s
        hi = int(np.searchsorted(r.pos, e, "left"))
        s1 = s - 1 if s > 0 else 0                        # the class before the tile's first position
        d1 = perbase_c(r.slice(min(lo, hi), hi), q, s1, e, diff=True)
        d = d1[s - s1:]
        eq, first = True, -1
        if got is not None:
            g = got[s:e]
            if not np.array_equal(g, d):
                eq, first = False, s + int(np.flatnonzero(g != d)[0])
        edges = np.arange(0, e - s, W)
        sums = np.add.reduceat(d.astype(np.int64), edges)
        mins = np.minimum.reduceat(d, edges)
        cls = np.where(d1 == 0, 0, np.where(d1 < mincov, 1, np.where((maxmean > 0) & (d1 >= maxmean), 3, 2))).astype(np.int8)
        c = cls[s - s1:]
        brk = np.empty(e - s, bool)
        brk[0] = True if s == 0 else (c[0] != cls[0])
        brk[1:] = c[1:] != c[:-1]
        brk[(-s) % step::step] = True
        st = np.flatnonzero(brk)
        return eq, first, sums, mins, st + s, c[st]

    with ThreadPoolExecutor(max_workers=max(1, workers)) as ex:
        res = list(ex.map(one, range(len(jobs))))
    bad = [x[1] for x in res if not x[0]]
    z64, z32, z8 = np.zeros(0, np.int64), np.zeros(0, np.int32), np.zeros(0, np.int8)
    return {"equal": not bad, "first_diff": min(bad) if bad else -1,
            "sums": np.concatenate([x[2] for x in res]) if res else z64,
            "mins": np.concatenate([x[3] for x in res]) if res else z32,
            "run_starts": np.concatenate([x[4] for x in res]) if res else z64,
            "run_cls": np.concatenate([x[5] for x in res]) if res else z8}


def chrom_start_end_c(line: bytes):
    chrom = ctypes.create_string_buffer(1024)
    s, e = ctypes.c_long(), ctypes.c_long()
    rc = lib().gdo_chrom_start_end(line, len(line), chrom, 1024, ctypes.byref(s), ctypes.byref(e))
    if rc != 0:
        raise ValueError("couldn't get region from line %r" % line)
    return chrom.value.decode(), s.value, e.value


def tiles_c(length: int, W: int):
    cap = lib().gdo_tiles(length, W, None, None, 0)
    s = np.zeros(cap, np.int64)
    e = np.zeros(cap, np.int64)
    lib().gdo_tiles(length, W, s.ctypes.data, e.ctypes.data, cap)
    return list(zip(s.tolist(), e.tolist()))


def callback_c(chrom: str, region_start: int, region_end: int, depth: np.ndarray,
               W: int, mincov: int, maxmean: int, depth_path: str, callable_path: str):
    d = np.ascontiguousarray(depth, dtype=np.int32)
    assert d.shape[0] == max(0, region_end - region_start)
    rc = lib().gdo_callback_append(chrom.encode(), region_start, region_end, d.ctypes.data,
                                   W, mincov, maxmean, depth_path.encode(),
                                   callable_path.encode())
    if rc != 0:
        raise OSError("gdo_callback_append failed")


def depth_run_oracle(contigs, reads_by_tid, W=250, Q=1, mincov=4, maxmean=0,
                     flag_mask=DEFAULT_FLAG_MASK, regions=None):
    """Whole `goleft depth` run through the oracle.

    contigs: list of (name, length).  reads_by_tid: {tid: Reads}.
    regions: None for whole-genome tiling (depth/depth.go:122-159), else a list
    of (chrom, start0, end) as --bed rows would give (depth.go:103-120).
    Returns (depth_bed_text, callable_bed_text) in genome / row order
    (what --ordered produces)."""
    import tempfile
    names = [c[0] for c in contigs]
    empty = Reads(np.zeros(0, np.int32), np.zeros(0, np.uint16), np.zeros(0, np.uint8),
                  np.zeros(1, np.uint32), np.zeros(0, np.uint32))
    jobs = []
    if regions is None:
        for tid, (name, length) in enumerate(contigs):
            for s, e in tiles_c(length, W):
                jobs.append((tid, name, s, e))
    else:
        for chrom, s, e in regions:
            jobs.append((names.index(chrom) if chrom in names else -1, chrom, s, e))
    with tempfile.TemporaryDirectory() as td:
        hd, ca = os.path.join(td, "d.bed"), os.path.join(td, "c.bed")
        open(hd, "w").close()
        open(ca, "w").close()
        for tid, name, s, e in jobs:
            r = reads_by_tid.get(tid, empty)
            d = perbase_c(r, Q, s, e, flag_mask)
            if tid >= 0:
                clen = contigs[tid][1]
                if e > clen:       # samtools prints nothing past the contig end
                    d[max(0, clen - s):] = 0
            callback_c(name, s, e, d, W, mincov, maxmean, hd, ca)
        return open(hd).read(), open(ca).read()


# ---------------------------------------------------------------------------
# depthwed (BASELINE.json config 4)
# ---------------------------------------------------------------------------
def depthwed_name(path: str) -> str:
    """depthwed/depthwed.go:37-46 getNameFromFile: basename, then ONE pass that strips
    a trailing .gz, then .bed, then .depth (in that order, each at most once)."""
    n = path.split("/")[-1]
    for suff in (".gz", ".bed", ".depth"):
        if n.endswith(suff):
            n = n[:-len(suff)]
    return n.rstrip("\n") if n.endswith("\n") else n


def depthwed_cell(mean_text: str) -> int:
    """depthwed/depthwed.go:96,:103: int(0.5 + ParseFloat(tok))."""
    return int(0.5 + float(mean_text))


def depthwed_cells_contig(sums, length: int, W: int, size: int):
    """One sample's column of the depthwed matrix for ONE contig, from its W-window sums -- the chain the reference
    runs through text, restated on arrays so that a whole chromosome (a million windows) takes seconds:
    goleft depth prints "%.4g" of float64(sum) / float64(window length) per W-window (depth/depth.go:181-189, :301;
    the last window of a contig is shorter); depthwed parses it back and adds int(0.5 + x) (depthwed/depthwed.go:96,
    :103) over the rows of a group, a group running from the current row until its span reaches `size`
    or the contig's rows run out (depthwed/depthwed.go:126: the loop also ends when the next row of the first file is on
    another chromosome -- or absent -- so the last, shorter group of a contig IS written; `eof` only ends a group that
    has read nothing).  -> (cells int64[rows], starts int64[rows], ends int64[rows]);
    tests/test_depthwed.py holds it against depthwed_py, the line-by-line restatement."""
    sums = np.asarray(sums, np.int64)
    nw = (length + W - 1) // W
    assert sums.shape[0] == nw
    ends = np.minimum((np.arange(nw, dtype=np.int64) + 1) * W, length)
    starts = np.arange(nw, dtype=np.int64) * W
    lens = ends - starts
    cell = np.fromiter((depthwed_cell("%.4g" % (0.0 if s == 0 else float(s) / float(l)))
                        for s, l in zip(sums.tolist(), lens.tolist())), np.int64, nw)
    out_c, out_s, out_e = [], [], []
    k = 0
    while k < nw:
        j = k
        while j + 1 < nw and ends[j] - starts[k] < size:
            j += 1                                   # rows k..j: the span reaches `size` with row j, or the rows run out
        out_c.append(int(cell[k:j + 1].sum()))
        out_s.append(int(starts[k]))
        out_e.append(int(ends[j]))
        k = j + 1
    return np.asarray(out_c, np.int64), np.asarray(out_s, np.int64), np.asarray(out_e, np.int64)


def depthwed_py(beds, names, size: int) -> str:
    """Line-by-line restatement of depthwed/depthwed.go:48-157 (`run` + `next`).

    beds: list of depth.bed texts (one per sample, same number of records);
    names: the column names (getNameFromFile of each path).  Returns the text
    the reference writes to stdout.  Records are grouped from the current row
    until the span reaches `size` or the next row of the FIRST file is on
    another chromosome (:126); each constituent row adds int(0.5 + mean)."""
    rows = [[ln for ln in b.split("\n") if ln != ""] for b in beds]
    ptr = 0
    n0 = len(rows[0])
    out = ["\t".join(["#chrom", "start", "end"] + list(names))]

    def next_chrom():
        return rows[0][ptr].split("\t")[0] if ptr < n0 else ""

    while True:
        # next(): depthwed.go:117-157
        depths = [None] * len(rows)
        eof = False
        k = 0
        chrom = next_chrom()
        span0 = 0
        while (not eof) and span0 < size and chrom == next_chrom():
            for i, rr in enumerate(rows):
                if ptr >= len(rr):
                    if i > 0 and not eof:
                        raise RuntimeError("not all files have same number of records")
                    eof = True
                    continue
                toks = rr[ptr].split("\t")
                c, s, e, d = toks[0], int(toks[1]), int(toks[2]), depthwed_cell(toks[3])
                if k == 0:
                    if c != chrom:
                        raise RuntimeError("got unexpected chromosome")
                    depths[i] = [c, s, e, d]
                else:
                    depths[i][2] = e
                    depths[i][3] += d
            if not eof:
                ptr += 1
                span0 = depths[0][2] - depths[0][1]
            k += 1
        if eof:
            break
        out.append("%s\t%d\t%d" % (depths[0][0], depths[0][1], depths[0][2]) +
                   "".join("\t%d" % d[3] for d in depths))
    return "\n".join(out) + "\n"
