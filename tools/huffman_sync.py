#!/usr/bin/env python3
"""No GPU: how quickly a DEFLATE decoder that starts at a WRONG bit offset falls into step with the true symbol sequence --
the number a wave-per-member inflate (the member's window in LDS, 64 lanes decoding one member
speculatively from 64 bit offsets and stitching) stands or falls with.

    python tools/huffman_sync.py x.bam [--members 24] [--starts 400] [--skip 40]

For members of the BAM: the true symbol boundaries of every Huffman block (a small DEFLATE parser, checked against zlib's
output length), then, from random bit offsets inside a block and with THAT block's tables (what a lane would have after
the header has been read once), symbols are decoded until the position is a true boundary.  Reported: the distribution of
bits and symbols until then, the share of starts that ran into an invalid code or the end of the block first, and what it
means for a wave: sequential symbol steps per member with 64 lanes against one."""
import argparse
import json
import random
import struct
import sys
import zlib

LEN_BASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
LEN_EXTRA = [0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0]
DIST_EXTRA = [0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13]
ORDER = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]


def build(lens):
    """canonical code -> {(length, code): symbol}"""
    count = [0] * 16
    for l in lens:
        count[l] += 1
    count[0] = 0
    code, nxt = 0, [0] * 16
    for b in range(1, 16):
        code = (code + count[b - 1]) << 1
        nxt[b] = code
    table = {}
    for s, l in enumerate(lens):
        if l:
            table[(l, nxt[l])] = s
            nxt[l] += 1
    return table


class Bits:
    def __init__(self, data):
        self.d, self.n = data, len(data) * 8

    def bit(self, pos):
        return (self.d[pos >> 3] >> (pos & 7)) & 1 if pos < self.n else 0

    def bits(self, pos, n):                                  # n bits, LSB first
        v = 0
        for k in range(n):
            v |= self.bit(pos + k) << k
        return v

    def sym(self, pos, table):                               # (symbol, new position) or (None, pos): no code / out of data
        code = 0
        for l in range(1, 16):
            if pos + l > self.n:
                return None, pos
            code = (code << 1) | self.bit(pos + l - 1)
            s = table.get((l, code))
            if s is not None:
                return s, pos + l
        return None, pos


def step(bs, pos, lit, dist):
    """one lit/len symbol (with its extra bits and distance): (kind, new position, bytes produced); kind None: invalid"""
    s, p = bs.sym(pos, lit)
    if s is None:
        return None, pos, 0
    if s < 256:
        return "lit", p, 1
    if s == 256:
        return "eob", p, 0
    if s > 285:
        return None, pos, 0
    ls = s - 257
    if p + LEN_EXTRA[ls] > bs.n:
        return None, pos, 0
    n = LEN_BASE[ls] + bs.bits(p, LEN_EXTRA[ls])
    p += LEN_EXTRA[ls]
    d, p2 = bs.sym(p, dist)
    if d is None or d > 29 or p2 + DIST_EXTRA[d] > bs.n:
        return None, pos, 0
    return "match", p2 + DIST_EXTRA[d], n


def parse_member(payload):
    """blocks: [(lit table, dist table, [bit position of every lit/len symbol], end position)] of the Huffman blocks"""
    bs = Bits(payload)
    pos, blocks, total = 0, [], 0
    while True:
        final = bs.bit(pos)
        typ = bs.bits(pos + 1, 2)
        pos += 3
        if typ == 0:
            pos = (pos + 7) & ~7
            n = bs.bits(pos, 16)
            pos += 32 + 8 * n
            total += n
        else:
            if typ == 1:
                lit = build([8] * 144 + [9] * 112 + [7] * 24 + [8] * 8)
                dist = build([5] * 30)
            else:
                nl, nd, nc = bs.bits(pos, 5) + 257, bs.bits(pos + 5, 5) + 1, bs.bits(pos + 10, 4) + 4
                pos += 14
                cl = [0] * 19
                for k in range(nc):
                    cl[ORDER[k]] = bs.bits(pos, 3)
                    pos += 3
                ct = build(cl)
                lens = []
                while len(lens) < nl + nd:
                    s, pos = bs.sym(pos, ct)
                    if s < 16:
                        lens.append(s)
                    elif s == 16:
                        lens += [lens[-1]] * (3 + bs.bits(pos, 2)); pos += 2
                    elif s == 17:
                        lens += [0] * (3 + bs.bits(pos, 3)); pos += 3
                    else:
                        lens += [0] * (11 + bs.bits(pos, 7)); pos += 7
                lit, dist = build(lens[:nl]), build(lens[nl:])
            starts = []
            while True:
                starts.append(pos)
                kind, pos, n = step(bs, pos, lit, dist)
                assert kind is not None
                total += n
                if kind == "eob":
                    break
            blocks.append((lit, dist, starts, pos))
        if final:
            return bs, blocks, total


def members(path):
    data = open(path, "rb").read()
    off = 0
    while off < len(data):
        xlen, = struct.unpack_from("<H", data, off + 10)
        bsize, = struct.unpack_from("<H", data, off + 16)
        yield data[off + 12 + xlen:off + bsize + 1 - 8], struct.unpack_from("<I", data, off + bsize + 1 - 4)[0]
        off += bsize + 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("bam")
    ap.add_argument("--members", type=int, default=24)
    ap.add_argument("--starts", type=int, default=400, help="random wrong bit offsets per member")
    ap.add_argument("--skip", type=int, default=40, help="take every n-th member")
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    rng = random.Random(a.seed)
    bits_to_sync, syms_to_sync, failed, n_symbols, n_blocks, n_members = [], [], 0, [], 0, 0
    for i, (payload, isize) in enumerate(members(a.bam)):
        if i % a.skip or isize < 1000:
            continue
        bs, blocks, total = parse_member(payload)
        assert total == isize == len(zlib.decompress(payload, -15)), (total, isize)
        n_members += 1
        n_blocks += len(blocks)
        n_symbols.append(sum(len(b[2]) for b in blocks))
        for _ in range(a.starts):
            lit, dist, starts, end = rng.choice(blocks)
            true = set(starts)
            p0 = rng.randrange(starts[0], starts[-1])
            if p0 in true:
                continue
            pos, k = p0, 0
            while pos not in true:
                kind, pos, _n = step(bs, pos, lit, dist)
                k += 1
                if kind is None or kind == "eob" or pos >= end:
                    k = -1
                    break
            if k < 0:
                failed += 1
            else:
                bits_to_sync.append(pos - p0)
                syms_to_sync.append(k)
        if n_members >= a.members:
            break
    bits_to_sync.sort(); syms_to_sync.sort()
    q = lambda v, f: v[min(len(v) - 1, int(f * len(v)))]
    mean_syms = sum(n_symbols) / len(n_symbols)
    out = {"members": n_members, "huffman_blocks_per_member": n_blocks / n_members, "symbols_per_member": mean_syms,
           "wrong_starts": len(bits_to_sync) + failed,
           "ran_into_an_invalid_code_or_the_block_end_first": failed / max(1, len(bits_to_sync) + failed),
           "symbols_until_in_step": {"median": q(syms_to_sync, 0.5), "p90": q(syms_to_sync, 0.9), "p99": q(syms_to_sync, 0.99), "max": syms_to_sync[-1]},
           "bits_until_in_step": {"median": q(bits_to_sync, 0.5), "p90": q(bits_to_sync, 0.9), "p99": q(bits_to_sync, 0.99), "max": bits_to_sync[-1]}}
    # 64 lanes on one member: every lane decodes its 1/64 of the symbols plus what its right neighbour needs to fall into step
    # (p99: a lane that has not is decoded again by its neighbour); one lane: all of them
    out["sequential_symbol_steps_per_member"] = {"one_lane": mean_syms, "64_lanes_speculative": mean_syms / 64 + q(syms_to_sync, 0.99)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
