#!/usr/bin/env python3
"""What the lanes of the inflate kernel do, iteration by iteration, on the members of a BAM file -- from the kernel's own
source run under the host emulation (tests/emul/inflate_stats.cpp), no GPU needed.  The kernel's time on the device is
(iterations of the longest lane of every wave) x (cost of an iteration) + (runs of the block-header path) x (its cost);
this prints the first factors of both products and how much of a wave's lane-iterations is useful work.
    python tools/inflate_model.py FILE.bam [--waves 40] [--skip 1]          (FILE.bam: e.g. from goleft_amd/synth-bam)"""
import argparse
import ctypes as C
import json
import os
import struct
import subprocess
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


class Stats(C.Structure):
    _fields_ = [("iters", C.c_uint64), ("lane_mode", C.c_uint64 * 4), ("hdr_runs", C.c_uint64), ("hdr_lanes", C.c_uint64),
                ("chunk_mem", C.c_uint64), ("chunk_ring", C.c_uint64), ("win_refill", C.c_uint64), ("mem_iters", C.c_uint64),
                ("src_hit", (C.c_uint64 * 3) * 3), ("mem_iters_after", (C.c_uint64 * 3) * 3), ("dist_le", C.c_uint64 * 8)]


def lib():
    src = os.path.join(ROOT, "tests", "emul", "inflate_stats.cpp")
    so = os.path.join(ROOT, "tests", "emul", "inflate_stats.so")
    deps = [src, os.path.join(ROOT, "goleft_amd", "csrc", "gd_inflate.hpp"), os.path.join(ROOT, "tests", "emul", "emul_machine.hpp")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call([CLANG, "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so, src])
    l = C.CDLL(so)
    l.emul_inflate_stats.argtypes = [C.c_void_p] * 7 + [C.c_uint32, C.c_uint32, C.POINTER(Stats)]
    return l


def members(path, first, count):
    """(payload bytes, isize) of members [first, first + count) of a BGZF file"""
    out = []
    with open(path, "rb") as fh:
        k = 0
        while len(out) < count:
            h = fh.read(18)
            if len(h) < 18:
                break
            bsize, = struct.unpack_from("<H", h, 16)
            body = fh.read(bsize + 1 - 18)
            if k >= first:
                out.append((body[:-8], struct.unpack_from("<I", body, len(body) - 4)[0]))
            k += 1
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("bam")
    ap.add_argument("--waves", type=int, default=40)
    ap.add_argument("--skip", type=int, default=1, help="members to skip at the start of the file (the header's)")
    ap.add_argument("--json", action="store_true")
    a = ap.parse_args()
    l = lib()
    ms = [m for m in members(a.bam, a.skip, a.waves * 64) if m[1] > 0]
    n = len(ms) // 64 * 64
    ms = ms[:n]
    in_off = np.zeros(n, np.uint64); in_len = np.zeros(n, np.uint32); out_off = np.zeros(n, np.uint64); out_len = np.zeros(n, np.uint32)
    p = q = 0
    for i, (c, isz) in enumerate(ms):
        in_off[i] = p; in_len[i] = len(c); out_off[i] = q; out_len[i] = isz
        p += len(c) + 8; q += isz
    comp = np.zeros(p + 256, np.uint8)
    for i, (c, _) in enumerate(ms):
        comp[int(in_off[i]):int(in_off[i]) + len(c)] = np.frombuffer(c, np.uint8)
    out = np.zeros(q + 256, np.uint8)
    status = np.full(n, 99, np.uint32)
    tot = Stats()
    per_wave = []
    for b in range(n // 64):
        st = Stats()
        l.emul_inflate_stats(comp.ctypes.data, in_off.ctypes.data, in_len.ctypes.data, out_off.ctypes.data, out_len.ctypes.data,
                             out.ctypes.data, status.ctypes.data, n, b, C.byref(st))
        per_wave.append(st.iters)
        for f, _ in Stats._fields_:
            if f == "lane_mode":
                for k in range(4):
                    tot.lane_mode[k] += st.lane_mode[k]
            elif f == "dist_le":
                for k in range(8):
                    tot.dist_le[k] += st.dist_le[k]
            elif f in ("src_hit", "mem_iters_after"):
                for z in range(3):
                    for w in range(3):
                        getattr(tot, f)[z][w] += getattr(st, f)[z][w]
            else:
                setattr(tot, f, getattr(tot, f) + getattr(st, f))
    assert (status[:n] == 0).all(), status[:n][status[:n] != 0][:8]
    for i, (c, isz) in enumerate(ms[:64]):                                 # (and the bytes are right)
        assert out[int(out_off[i]):int(out_off[i]) + isz].tobytes() == zlib.decompress(c, -15)
    lane_iters = sum(tot.lane_mode)
    res = {
        "members": n, "waves": n // 64, "output_bytes": int(q), "input_bytes": int(sum(len(c) for c, _ in ms)),
        "iterations_per_wave": {"mean": float(np.mean(per_wave)), "min": int(min(per_wave)), "max": int(max(per_wave))},
        "output_bytes_per_lane_iteration": q / lane_iters,
        "lane_iterations": {"decode_a_symbol": tot.lane_mode[0] / lane_iters, "copy_a_chunk": tot.lane_mode[1] / lane_iters,
                            "wait_for_a_header_run": tot.lane_mode[2] / lane_iters, "finished_waiting_for_the_wave": tot.lane_mode[3] / lane_iters},
        "header_runs_per_wave": tot.hdr_runs / (n // 64), "lanes_per_header_run": tot.hdr_lanes / max(1, tot.hdr_runs),
        "chunks_from_memory_per_member": tot.chunk_mem / n, "chunks_from_the_ring_per_member": tot.chunk_ring / n,
        "input_slots_loaded_per_member": tot.win_refill / n,
        "iterations_with_a_chunk_load_from_memory": tot.mem_iters / tot.iters,
        # how far back the sources of the chunk loads from memory lie (cumulative shares)
        "chunk_loads_from_memory_with_distance_at_most": {str(l): tot.dist_le[k] / max(1, tot.chunk_mem)
                                                          for k, l in enumerate((128, 256, 512, 1024, 2048, 4096, 8192, 32768))},
        # a per-lane cache of the aligned block(s) the last chunk sources were loaded from: the share of the loads from memory
        # it would serve, and the share of iterations in which some lane would still load
        "source_block_cache": {"%d B x %d" % (64 << z, 1 << w): {"loads_served": tot.src_hit[z][w] / max(1, tot.chunk_mem),
                                                                  "iterations_still_loading": tot.mem_iters_after[z][w] / tot.iters}
                               for z in range(3) for w in range(3)},
    }
    print(json.dumps(res, indent=None if a.json else 1))


if __name__ == "__main__":
    main()
