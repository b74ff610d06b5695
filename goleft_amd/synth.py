"""Synthetic decoded-record streams, bit-identical in
numpy (host, tests / CPU baseline) and torch (device, bench.py).

Counter-based: value = mix64(counter, seed, stream) with the splitmix64
finaliser, so any slice can be regenerated anywhere.  Starts are drawn
uniformly and then sorted; every other attribute is a function of the sorted
rank, so the stream does not depend on the sort implementation.

Short-read model (configs 2/3: 150 bp, 30x):
  92 % 150M | 5 % kS(150-k)M, k in 1..30 | 2 % aM dD (150-a)M, d in 1..10 |
  1 % aM iI (150-a-i)M, i in 1..10;
  flags: 5 % DUP, 0.1 % each SECONDARY / QCFAIL / UNMAP, 0.5 % SUPPLEMENTARY
  (counted by samtools depth), random strand; MAPQ 0 for 1 %, else 60.
"""
from __future__ import annotations

import os

import numpy as np

READ_LEN = 150
_M64 = (1 << 64) - 1
_C1 = 0xBF58476D1CE4E5B9
_C2 = 0x94D049BB133111EB
_G = 0x9E3779B97F4A7C15
_S = 0xD1B54A32D192ED03
_T = 0x8CB92BA72F3D8DD7

# hg19 primary contig lengths (chr1..22, X, Y), sum 3 095 677 412
HG19_LENGTHS = [249250621, 243199373, 198022430, 191154276, 180915260, 171115067, 159138663,
                146364022, 141213431, 135534747, 135006516, 133851895, 115169878, 107349540,
                102531392, 90354753, 81195210, 78077248, 59128983, 63025520, 48129895, 51304566,
                155270560, 59373566]
HG19_NAMES = ["chr%d" % i for i in range(1, 23)] + ["chrX", "chrY"]
CHR20_LEN = 63025520


def n_reads_for(length: int, coverage: float = 30.0, read_len: int = READ_LEN) -> int:
    return int(round(length * coverage / read_len))


# ---------------------------------------------------------------------------
# numpy backend
# ---------------------------------------------------------------------------
def _np_mix(counter: np.ndarray, seed: int, stream: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (counter.astype(np.uint64) + np.uint64(1)) * np.uint64(_G)
        x = x + np.uint64((seed * _S + stream * _T) & _M64)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(_C1)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(_C2)
        x = x ^ (x >> np.uint64(31))
    return x


def _np_u(counter, seed, stream, mod):
    return ((_np_mix(counter, seed, stream) >> np.uint64(11)) % np.uint64(mod)).astype(np.int64)


def short_reads_numpy(length: int, n: int, seed: int):
    """Returns (pos i32, flag u16, mapq u8, cigar_off u32, cigar u32)."""
    idx = np.arange(n, dtype=np.uint64)
    pos = np.sort(_np_u(idx, seed, 1, max(1, length - READ_LEN + 1))).astype(np.int32)
    kind_r = _np_u(idx, seed, 2, 10000)
    u3 = _np_mix(idx, seed, 3) >> np.uint64(11)
    u4 = _np_mix(idx, seed, 4) >> np.uint64(11)
    f5 = _np_u(idx, seed, 5, 1000)
    u6 = _np_mix(idx, seed, 6) >> np.uint64(11)
    return _assemble(np, pos, kind_r, u3.astype(np.int64), u4.astype(np.int64), f5,
                     u6.astype(np.int64))


# ---------------------------------------------------------------------------
# torch backend (same integer arithmetic on int64 with explicit logical shifts)
# ---------------------------------------------------------------------------
def _s64(c: int) -> int:
    c &= _M64
    return c - (1 << 64) if c >= (1 << 63) else c


def _t_lsr(x, s: int):
    return (x >> s) & ((1 << (64 - s)) - 1)


def _t_mix(counter, seed: int, stream: int):
    x = (counter + 1) * _s64(_G)
    x = x + _s64(seed * _S + stream * _T)
    x = (x ^ _t_lsr(x, 30)) * _s64(_C1)
    x = (x ^ _t_lsr(x, 27)) * _s64(_C2)
    x = x ^ _t_lsr(x, 31)
    return x


def short_reads_torch(length: int, n: int, seed: int, device):
    """Device twin of short_reads_numpy; returns torch tensors
    (pos i32, flag i16 [bit pattern of u16], mapq u8, cigar_off i32, cigar i32)."""
    import torch
    idx = torch.arange(n, dtype=torch.int64, device=device)
    u1 = _t_lsr(_t_mix(idx, seed, 1), 11) % max(1, length - READ_LEN + 1)
    pos = torch.sort(u1).values.to(torch.int32)
    del u1
    kind_r = _t_lsr(_t_mix(idx, seed, 2), 11) % 10000
    u3 = _t_lsr(_t_mix(idx, seed, 3), 11)
    u4 = _t_lsr(_t_mix(idx, seed, 4), 11)
    f5 = _t_lsr(_t_mix(idx, seed, 5), 11) % 1000
    u6 = _t_lsr(_t_mix(idx, seed, 6), 11)
    del idx
    return _assemble(torch, pos, kind_r, u3, u4, f5, u6)


# ---------------------------------------------------------------------------
# shared assembly (xp is numpy or torch; only ops common to both are used)
# ---------------------------------------------------------------------------
def _assemble(xp, pos, kind_r, u3, u4, f5, u6):
    is_t = xp.__name__ == "torch"

    def where(c, a, b):
        return xp.where(c, a, b)

    def full_like(a, v):
        return xp.full_like(a, v)

    RL = READ_LEN
    kind = (kind_r >= 9200) * 1 + (kind_r >= 9700) * 1 + (kind_r >= 9900) * 1   # 0..3
    if os.environ.get("GOLEFT_SYNTH_PLAIN") == "1":          # MEASUREMENTS ONLY: every read `150M` (what a kernel's rare paths cost)
        kind = kind * 0
    elif os.environ.get("GOLEFT_SYNTH_PLAIN") == "2":        # ... `150M` or `kS(150-k)M`: no read of three ops
        kind = kind * (kind <= 1)
    k_clip = 1 + u3 % 30
    a_del = 20 + u3 % 111
    d_len = 1 + u4 % 10
    a_ins = 20 + u3 % 101
    i_len = 1 + u4 % 10
    M, I, D, S = 0, 1, 2, 4
    op0 = where(kind == 0, full_like(kind, (RL << 4) | M),
                where(kind == 1, (k_clip << 4) | S,
                      where(kind == 2, (a_del << 4) | M, (a_ins << 4) | M)))
    op1 = where(kind == 1, ((RL - k_clip) << 4) | M,
                where(kind == 2, (d_len << 4) | D, (i_len << 4) | I))
    op2 = where(kind == 2, ((RL - a_del) << 4) | M, ((RL - a_ins - i_len) << 4) | M)
    nops = 1 + (kind >= 1) * 1 + (kind >= 2) * 1
    n = pos.shape[0]
    if is_t:
        import torch
        off = torch.zeros(n + 1, dtype=torch.int64, device=pos.device)
        torch.cumsum(nops, 0, out=off[1:])
        m = int(off[-1].item())
        cigar = torch.zeros(m, dtype=torch.int32, device=pos.device)
        cigar[off[:-1]] = op0.to(torch.int32)
        s1 = kind >= 1
        cigar[off[:-1][s1] + 1] = op1[s1].to(torch.int32)
        s2 = kind >= 2
        cigar[off[:-1][s2] + 2] = op2[s2].to(torch.int32)
        cigar_off = off.to(torch.int32)
    else:
        off = np.zeros(n + 1, np.int64)
        np.cumsum(nops, out=off[1:])
        cigar = np.zeros(int(off[-1]), np.uint32)
        cigar[off[:-1]] = op0.astype(np.uint32)
        s1 = kind >= 1
        cigar[off[:-1][s1] + 1] = op1[s1].astype(np.uint32)
        s2 = kind >= 2
        cigar[off[:-1][s2] + 2] = op2[s2].astype(np.uint32)
        cigar_off = off.astype(np.uint32)

    flag = full_like(f5, 0x1 | 0x2) + (u6 & 1) * 0x10 + where(((u6 >> 1) & 1) == 1,
                                                              full_like(f5, 0x40),
                                                              full_like(f5, 0x80))
    flag = flag + (f5 < 50) * 0x400 + (f5 == 50) * 0x100 + (f5 == 51) * 0x200 + (f5 == 52) * 0x4
    flag = flag + ((f5 >= 53) & (f5 < 58)) * 0x800
    mapq = where(((u6 >> 8) % 100) == 0, full_like(f5, 0), full_like(f5, 60))
    if is_t:
        import torch
        return (pos, flag.to(torch.int16), mapq.to(torch.uint8), cigar_off, cigar)
    return (pos, flag.astype(np.uint16), mapq.astype(np.uint8), cigar_off, cigar)


def algorithmic_bytes(n_reads: int, n_ops: int, n_bases: int, n_windows: int, raw: bool = False) -> int:
    """4*reads(pos) + 4*reads(CSR offsets, read on device)
    + 4*ops + 4*bases (int32 per-base write) + 8*windows (int64 sums).  raw: the kernel reads the
    records as they arrived -- flag (2) and MAPQ (1) per read too, which the canonical record word
    folds into the 4 bytes of the CSR offset."""
    return (11 if raw else 8) * n_reads + 4 * n_ops + 4 * n_bases + 8 * n_windows


# ---------------------------------------------------------------------------
# Long-read model (BASELINE.json config 5: 20x ONT, N50 ~ 20 kb, indel-heavy)
# ---------------------------------------------------------------------------
# Integer-only, so numpy and torch agree bit for bit:
#   * aligned length of a read = entry (u % 1024) of a 1024-point quantile table
#     of the log-normal(mu = ln 20000 - sigma^2, sigma = 0.8) (built once with
#     numpy on the host for both backends), i.e. length-weighted median ~20 kb;
#   * CIGAR = S, then k alternating runs  M (1..49, mean 25)  and  I or D (1:1;
#     length 1..5 with P = .60 .24 .10 .04 .02, mean 1.64), then S;
#     k = max(1, aligned length // 27) match runs => ~1 op per 13 aligned bases;
#   * flags: 10 % SUPPLEMENTARY (counted by samtools depth), 3 % SECONDARY
#     (dropped); MAPQ 0 for 2 %, else 60; starts uniform, sorted; every other
#     attribute is a function of the sorted rank / the global op index.
ONT_SIGMA = 0.8
ONT_MEDIAN_W = 20000.0
_ONT_TABLE = None


def ont_length_table() -> np.ndarray:
    """1024 quantiles of the aligned-length distribution (int64, >= 200)."""
    global _ONT_TABLE
    if _ONT_TABLE is None:
        from statistics import NormalDist
        mu = np.log(ONT_MEDIAN_W) - ONT_SIGMA ** 2
        q = (np.arange(1024) + 0.5) / 1024.0
        z = np.array([NormalDist().inv_cdf(float(x)) for x in q])
        _ONT_TABLE = np.maximum(200, np.exp(mu + ONT_SIGMA * z)).astype(np.int64)
    return _ONT_TABLE


def ont_mean_length() -> float:
    t = ont_length_table()
    k = np.maximum(1, t // 27)
    return float((k * 25 + (k - 1) * 0.5 * 1.64).mean())     # M bases + D bases per read


def n_ont_reads_for(length: int, coverage: float = 20.0) -> int:
    return max(1, int(round(length * coverage / ont_mean_length())))


def ont_reads_numpy(length: int, n: int, seed: int):
    """Returns (pos i32, flag u16, mapq u8, cigar_off u32, cigar u32)."""
    idx = np.arange(n, dtype=np.uint64)
    pos = np.sort(_np_u(idx, seed, 11, max(1, length))).astype(np.int32)
    tab = ont_length_table()
    alen = tab[_np_u(idx, seed, 12, 1024)]
    k = np.maximum(1, alen // 27)
    nops = 2 * k + 1                                   # S (M x)* M S  with k M runs
    off = np.zeros(n + 1, np.int64)
    np.cumsum(nops, out=off[1:])
    m = int(off[-1])
    rid = np.repeat(np.arange(n, dtype=np.int64), nops)
    j = np.arange(m, dtype=np.int64)
    i = j - off[rid]
    last = nops[rid] - 1
    u = (_np_mix(j.astype(np.uint64), seed, 13) >> np.uint64(11)).astype(np.int64)
    cigar = _ont_ops(np, i, last, u).astype(np.uint32)
    f5 = _np_u(idx, seed, 14, 1000)
    flag = (np.where(f5 < 100, 0x800, 0) + np.where((f5 >= 100) & (f5 < 130), 0x100, 0) +
            (f5 % 2) * 0x10).astype(np.uint16)
    mapq = np.where(_np_u(idx, seed, 15, 100) < 2, 0, 60).astype(np.uint8)
    return pos, flag, mapq, off.astype(np.uint32), cigar


def ont_reads_torch(length: int, n: int, seed: int, device):
    """Device twin of ont_reads_numpy (pos i32, flag i16, mapq u8, cigar_off i32, cigar i32)."""
    import torch
    idx = torch.arange(n, dtype=torch.int64, device=device)
    pos = torch.sort(_t_lsr(_t_mix(idx, seed, 11), 11) % max(1, length)).values.to(torch.int32)
    tab = torch.from_numpy(ont_length_table()).to(device)
    alen = tab[_t_lsr(_t_mix(idx, seed, 12), 11) % 1024]
    k = torch.clamp(alen // 27, min=1)
    nops = 2 * k + 1
    off = torch.zeros(n + 1, dtype=torch.int64, device=device)
    torch.cumsum(nops, 0, out=off[1:])
    m = int(off[-1].item())
    rid = torch.repeat_interleave(torch.arange(n, dtype=torch.int64, device=device), nops, output_size=m)
    j = torch.arange(m, dtype=torch.int64, device=device)
    i = j - off[rid]
    last = nops[rid] - 1
    del rid
    u = _t_lsr(_t_mix(j, seed, 13), 11)
    del j
    cigar = _ont_ops(torch, i, last, u).to(torch.int32)
    del i, last, u
    f5 = _t_lsr(_t_mix(idx, seed, 14), 11) % 1000
    flag = (torch.where(f5 < 100, 0x800, 0) + torch.where((f5 >= 100) & (f5 < 130), 0x100, 0) +
            (f5 % 2) * 0x10).to(torch.int16)
    mapq = torch.where(_t_lsr(_t_mix(idx, seed, 15), 11) % 100 < 2, 0, 60).to(torch.uint8)
    return pos, flag, mapq, off.to(torch.int32), cigar


def _ont_ops(xp, i, last, u):
    """BAM-encoded op for local index i of a read with `last`+1 ops; u = 53 random bits."""
    M, I, D, S = 0, 1, 2, 4
    clip = (i == 0) | (i == last)
    is_m = (i % 2) == 1
    m_len = 1 + u % 49
    g = (u >> 8) % 100
    id_len = 1 + (g >= 60) * 1 + (g >= 84) * 1 + (g >= 94) * 1 + (g >= 98) * 1
    id_op = xp.where(((u >> 20) & 1) == 1, xp.full_like(i, I), xp.full_like(i, D))
    s_len = 1 + (u >> 24) % 200
    return xp.where(clip, (s_len << 4) | S, xp.where(is_m, (m_len << 4) | M, (id_len << 4) | id_op))
