#!/usr/bin/env python3
"""Device BGZF inflate (gd_inflate_bgzf) against zlib on a synthetic BAM: correctness + kernel time."""
import os
import struct
import subprocess
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from goleft_amd.engine import DepthEngine, K_INFLATE

length = sys.argv[1] if len(sys.argv) > 1 else "10000000"      # one contig length, or several separated by commas
pads = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]      # GD_OPT_INFLATE_LDS_PAD values to time
# GD_OPT_INFLATE_KERNEL values to time: 0 = a lane per member (the default), 1 = a workgroup per member (round 6)
probes = [int(x) for x in os.environ.get("INFLATE_BENCH_KERNELS", "0,1").split(",")]
path = "/tmp/gd_inflate_test.bam"
subprocess.check_call([os.path.join(ROOT, "goleft_amd", "synth-bam"), path, "chr20", length, "30", "20"],
                      stdout=subprocess.DEVNULL)
data = open(path, "rb").read()


def zlib_bgzf(raw):
    """The members of a BGZF byte string inflated with Python's zlib (the yardstick)."""
    out, off = [], 0
    while off < len(raw):
        xlen, = struct.unpack_from("<H", raw, off + 10)
        bsize, = struct.unpack_from("<H", raw, off + 16)          # synth-bam writes BC as the only subfield
        out.append(zlib.decompress(raw[off + 12 + xlen:off + bsize + 1 - 8], -15))
        off += bsize + 1
    return b"".join(out)


t0 = time.perf_counter()
want = None if os.environ.get("INFLATE_BENCH_NO_ZLIB") else zlib_bgzf(data)      # (counter passes: the yardstick ran in the trace pass)
t_cpu = time.perf_counter() - t0
with DepthEngine(0) as eng:
    eng.set_profiling(True)
    eng.inflate_bgzf(data[:1 << 20] if False else data)          # warm-up (allocations, code load)
    for pad, probe in [(a, b) for a in pads for b in probes]:
        eng.set_option(16, pad)                                  # GD_OPT_INFLATE_LDS_PAD
        eng.set_option(20, probe)                                # GD_OPT_INFLATE_KERNEL
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            got, status = eng.inflate_bgzf(data)
            t_all = time.perf_counter() - t0
            ms = eng.kernel_ms(K_INFLATE)
            best = ms if best is None else min(best, ms)
        ms = best
        print("lds pad %6d kernel %d: members %d, %.1f MB -> %.1f MB; status ok %s; equal %s" % (pad, probe, len(status), len(data) / 1e6, len(got) / 1e6,
                                                                                   bool((status == 0).all()), None if want is None else got == want))
        print("   kernel %.2f ms = %.2f GB/s of output (%.2f GB/s of BGZF); python zlib 1 thread %.2f s; call incl. H2D/D2H %.3f s"
              % (ms, len(got) / ms / 1e6, len(data) / ms / 1e6, t_cpu, t_all), flush=True)
        # a measurement build (-DGD_MEASURE, loaded through GOLEFT_DEPTH_SO): where the waves' cycles went
        import ctypes
        from goleft_amd import _lib
        dbg = getattr(_lib.load(), "gd_debug_inflate_sections", None) if os.environ.get("GOLEFT_DEPTH_SO") else None
        if dbg is not None:
            buf = (ctypes.c_ulonglong * 16)()
            dbg.restype = ctypes.c_int
            if dbg(buf) == 0 and buf[7]:
                names = ["top: exits, this iteration's loads, block headers", "ring / window reads, decode of three symbols", "wait for the two loads (vmcnt 0)",
                         "append the chunk, the symbol's branch", "window put, refill of the bit buffer", "ring write, 64-byte block stores", "plan the next chunk"]
                tot = float(sum(buf[k] for k in range(7)))
                print("   sections over %d wave-iterations (%.0f cycles each):" % (buf[7], tot / buf[7]))
                for k in range(7):
                    print("     %5.1f %%  %7.0f cycles  %s" % (100.0 * buf[k] / tot, buf[k] / buf[7], names[k]))
            if buf[15]:
                names = ["block header (stage, code lengths)", "tables", "pass A (speculative run + restarts)", "scan, bitmap clear", "pass B1 (literals, pieces)",
                         "pass B2 (pieces in output order)", "store, status"]
                tot = float(sum(buf[8 + k] for k in range(7)))
                print("   workgroup-per-member kernel, %d members (%.0f cycles each, first wave's clock):" % (buf[15], tot / buf[15]))
                for k in range(7):
                    print("     %5.1f %%  %8.0f cycles  %s" % (100.0 * buf[8 + k] / tot, buf[8 + k] / buf[15], names[k]))
                b2 = (ctypes.c_ulonglong * 8)()
                f2 = getattr(_lib.load(), "gd_debug_inflate_b2", None)
                if f2 is not None and f2(b2) == 0 and b2[4]:
                    n = float(buf[15])
                    print("     pass B2 per member: %.1f windows, %.1f batches, %.1f rounds; cycles: bitmap -> piece starts %.0f, batch set-up %.0f (%.0f each), rounds %.0f (%.0f each)"
                          % (b2[3] / n, b2[4] / n, b2[5] / n, b2[0] / n, b2[1] / n, b2[1] / max(1, b2[4]), b2[2] / n, b2[2] / max(1, b2[5])))
os.unlink(path)
