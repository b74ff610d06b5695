"""CPU: the device inflate kernels (goleft_amd/csrc/gd_inflate.hpp) compiled for the HOST and run lane by lane as
fibers (tests/emul/inflate_emul.cpp) against zlib -- the state machine, the table builder and the CRC kernel as the
product compiles them, on streams chosen for the rare paths: stored and fixed blocks, members shorter than the
16-byte register window, matches of every distance below 16, chunks that end at the member's end, several blocks per
member, damaged payloads.  The GPU tests compare the kernel on an MI355X with zlib (tests/test_gpu_bamdecode.py);
this is the same check where no GPU is needed, and tens of times as many streams."""
import ctypes as C
import os
import shutil
import subprocess
import sys
import zlib

import numpy as np
import pytest

from tests import helpers as H

EMUL_DIR = os.path.join(H.ROOT, "tests", "emul")
CLANG = next((p for p in ("/opt/rocm/lib/llvm/bin/clang++", shutil.which("clang++") or "") if p and os.path.exists(p)), None)

pytestmark = pytest.mark.skipif(CLANG is None, reason="the kernel source needs clang (ext_vector_type) to compile for the host")


def _lib():
    src = os.path.join(EMUL_DIR, "inflate_emul.cpp")
    so = os.path.join(EMUL_DIR, "inflate_emul.so")
    hdrs = [os.path.join(H.ROOT, "goleft_amd", "csrc", h) for h in ("gd_inflate.hpp",)] + [os.path.join(EMUL_DIR, "emul_machine.hpp")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(x) for x in [src] + hdrs):
        subprocess.check_call([CLANG, "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so, src])
    lib = C.CDLL(so)
    lib.emul_inflate.argtypes = [C.c_void_p] * 8 + [C.c_uint32]
    lib.emul_inflate.restype = C.c_int
    lib.emul_inflate_guarded.argtypes = [C.c_void_p, C.c_uint64] + [C.c_void_p] * 6 + [C.c_uint64, C.c_void_p, C.c_uint32]
    lib.emul_inflate_guarded.restype = C.c_int
    return lib


# GD_OPT_INFLATE_KERNEL under the emulation: 0 the lane-per-member kernel alone (the product's default), 1 the
# workgroup-per-member kernel (gd_inflate_wave.hpp) with the lane-per-member kernel behind it for the members it left.
# Every test of this file runs on the default; the ones marked `both_kernels` run on either.
@pytest.fixture(autouse=True)
def _kernel(request):
    k = getattr(request, "param", 0)
    _lib().emul_inflate_kernel(k)
    yield k
    _lib().emul_inflate_kernel(0)


both_kernels = pytest.mark.parametrize("_kernel", [0, 1], indirect=True, ids=["lane-per-member", "workgroup-per-member"])


def fallbacks() -> int:
    """members the last emulated launch of the workgroup-per-member kernel left to the lane-per-member kernel"""
    lib = _lib()
    lib.emul_inflate_fallbacks.restype = C.c_uint32
    return int(lib.emul_inflate_fallbacks())


def deflate(x: bytes, level=6, strategy=zlib.Z_DEFAULT_STRATEGY) -> bytes:
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
    return co.compress(x) + co.flush()


def emul_inflate(payloads, sizes, crcs=None, guarded=False):
    """payloads: raw deflate streams; sizes: what each is said to inflate to.  Laid out like BGZF members: eight bytes
    of trailer after every payload, the outputs back to back.  -> (bytes of every member, status[]).  guarded: both
    buffers end gd::INF_SLACK bytes before an inaccessible page (a stray access ends the process)."""
    lib = _lib()
    n = len(payloads)
    in_off = np.zeros(n, np.uint64); in_len = np.zeros(n, np.uint32)
    out_off = np.zeros(n, np.uint64); out_len = np.asarray(sizes, np.uint32)
    p = q = 0
    for i, c in enumerate(payloads):
        in_off[i] = p; in_len[i] = len(c); out_off[i] = q
        p += len(c) + 8
        q += int(out_len[i])
    comp = np.full(p + 256, 0xa5, np.uint8)                 # (the slack the device buffers have: gd::INF_SLACK)
    for i, c in enumerate(payloads):
        comp[int(in_off[i]):int(in_off[i]) + len(c)] = np.frombuffer(c, np.uint8)
    out = np.full(q + 256, 0xee, np.uint8)
    status = np.full(n, 99, np.uint32)
    crc = None if crcs is None else np.asarray(crcs, np.uint32)
    if guarded:
        rc = lib.emul_inflate_guarded(comp.ctypes.data, p, in_off.ctypes.data, in_len.ctypes.data, out_off.ctypes.data,
                                      out_len.ctypes.data, None if crc is None else crc.ctypes.data, out.ctypes.data, q,
                                      status.ctypes.data, n)
        assert rc == 0, "the kernel wrote past the last member (-2), or the two CRC kernels disagree (-3): %d" % rc
    else:
        rc = lib.emul_inflate(comp.ctypes.data, in_off.ctypes.data, in_len.ctypes.data, out_off.ctypes.data, out_len.ctypes.data,
                         None if crc is None else crc.ctypes.data, out.ctypes.data, status.ctypes.data, n)
        assert rc == 0, "the two CRC kernels disagree (-3): %d" % rc
    assert (out[q:] == 0xee).all(), "the kernel wrote past the last member"
    return [out[int(out_off[i]):int(out_off[i]) + int(out_len[i])].tobytes() for i in range(n)], status


def check(parts, how):
    payloads = [deflate(x, lv, st) for x, (lv, st) in zip(parts, how)]
    got, status = emul_inflate(payloads, [len(x) for x in parts], [zlib.crc32(x) & 0xffffffff for x in parts])
    for i, x in enumerate(parts):
        assert status[i] == 0 and got[i] == x, (i, int(status[i]), len(x), how[i],
                                                 next((k for k in range(len(x)) if got[i][k] != x[k]), None))


STRATEGIES = [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED]


def random_part(rng, n):
    kind = int(rng.integers(0, 7))
    if kind == 0:
        x = rng.integers(0, 256, n, dtype=np.uint8)
    elif kind == 1:
        x = rng.integers(0, 4, n, dtype=np.uint8) + 65
    elif kind == 2:
        x = np.repeat(rng.integers(0, 256, n // 50 + 1, dtype=np.uint8), 50)[:n]
    elif kind == 3:                                          # a period below 16: the chunks built from the register window
        d = int(rng.integers(1, 16))
        x = np.tile(rng.integers(0, 256, d, dtype=np.uint8), n // d + 1)[:n]
    elif kind == 4:                                          # a period of 16..40: chunks that read what the previous chunk wrote
        d = int(rng.integers(16, 41))
        x = np.tile(rng.integers(0, 256, d, dtype=np.uint8), n // d + 1)[:n]
    elif kind == 5:                                          # BAM-like: short random runs, quality-like plateaus
        x = np.concatenate([np.concatenate([rng.integers(0, 16, 30, dtype=np.uint8) * 17,
                                            np.full(int(rng.integers(1, 80)), int(rng.integers(30, 42)), np.uint8)])
                            for _ in range(n // 60 + 1)])[:n]
    else:
        x = np.zeros(n, np.uint8)
    return x.tobytes()


@both_kernels
def test_short_members_and_every_small_distance():
    """Members shorter than the 16-byte window, a member of every length 0..48, every period 1..40 at lengths that
    end a chunk exactly at, one before and one past the member's end."""
    rng = np.random.default_rng(1)
    parts, how = [], []
    for n in range(0, 49):
        parts.append(rng.integers(0, 256, n, dtype=np.uint8).tobytes()); how.append((6, zlib.Z_DEFAULT_STRATEGY))
    for d in range(1, 41):
        base = rng.integers(0, 256, d, dtype=np.uint8)
        for n in (d + 3, 16 + d, 31, 32, 33, 47, 48, 49, 258 + d, 300):
            parts.append(np.tile(base, n // d + 1)[:n].tobytes()); how.append((int(rng.integers(1, 10)), zlib.Z_DEFAULT_STRATEGY))
    check(parts, how)


@both_kernels
def test_stored_fixed_and_empty_blocks():
    rng = np.random.default_rng(2)
    parts, how = [], []
    for n in (0, 1, 15, 16, 17, 1000, 40000):
        x = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        parts += [x, x, x]
        how += [(0, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED), (1, zlib.Z_HUFFMAN_ONLY)]
    check(parts, how)
    # stored, dynamic and fixed blocks in ONE member (Z_FULL_FLUSH leaves an empty stored block between them)
    a, b, c = (rng.integers(0, 8, 3000, dtype=np.uint8).tobytes() for _ in range(3))
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    payload = co.compress(a) + co.flush(zlib.Z_FULL_FLUSH) + co.compress(b) + co.flush(zlib.Z_SYNC_FLUSH) + co.compress(c) + co.flush()
    got, status = emul_inflate([payload], [9000], [zlib.crc32(a + b + c) & 0xffffffff])
    assert status[0] == 0 and got[0] == a + b + c


@pytest.mark.parametrize("seed", range(3))
def test_random_members_of_every_kind(seed):
    """Two workgroups of small members (every payload kind x level x strategy) and a few of BGZF's full size (four or
    five blocks each: the lanes of a wave meet their headers at different iterations)."""
    rng = np.random.default_rng(100 + seed)
    parts = [random_part(rng, int(rng.integers(0, 5000))) for _ in range(120)]
    parts += [random_part(rng, int(rng.integers(30000, 65281))) for _ in range(8)]
    how = [(int(rng.integers(0, 10)), STRATEGIES[int(rng.integers(0, 5))]) for _ in parts]
    check(parts, how)


@both_kernels
def test_damaged_payloads_are_refused_or_inflate_to_what_zlib_makes_of_them():
    """Bit flips in the payload: the kernel reports an error (a decoder code, or 18 from the CRC kernel), or -- when the
    damaged stream is still a valid one -- delivers exactly zlib's bytes; it never writes outside the member (the guard
    bytes behind the output are checked by emul_inflate) and never runs away (the process would not return)."""
    rng = np.random.default_rng(7)
    good = [random_part(rng, int(rng.integers(200, 4000))) for _ in range(40)]
    payloads, sizes, crcs, want = [], [], [], []
    for x in good:
        c = bytearray(deflate(x, int(rng.integers(1, 10)), STRATEGIES[int(rng.integers(0, 5))]))
        for _ in range(int(rng.integers(1, 4))):
            c[int(rng.integers(0, len(c)))] ^= 1 << int(rng.integers(0, 8))
        try:
            d = zlib.decompressobj(-15)
            y = d.decompress(bytes(c)) + d.flush()
            ok = d.eof and len(y) == len(x)
        except zlib.error:
            y, ok = b"", False
        payloads.append(bytes(c)); sizes.append(len(x)); crcs.append(zlib.crc32(x) & 0xffffffff)
        want.append(y if ok else None)
    got, status = emul_inflate(payloads, sizes, crcs)
    n_refused = 0
    for i, x in enumerate(good):
        if status[i] == 0:
            assert got[i] == x                              # the CRC of the intact payload held: these are its bytes
        else:
            n_refused += 1
            if status[i] == 18:                             # inflated cleanly to other bytes: zlib agrees on which
                assert want[i] is not None and got[i] == want[i]
    assert n_refused >= 20


GUARDED_FUZZ = r"""
import sys, zlib
import numpy as np
sys.path.insert(0, %r)
from tests import test_inflate_emul as T
rng = np.random.default_rng(int(sys.argv[1]))
payloads, sizes = [], []
for k in range(192):
    x = T.random_part(rng, int(rng.integers(1, 3000)))
    c = bytearray(T.deflate(x, int(rng.integers(0, 10)), T.STRATEGIES[int(rng.integers(0, 5))]))
    mode = k %% 4
    if mode == 0:                                            # bit flips
        for _ in range(int(rng.integers(1, 6))):
            c[int(rng.integers(0, len(c)))] ^= 1 << int(rng.integers(0, 8))
    elif mode == 1:                                          # truncated: the stream runs into the trailer and beyond
        c = c[:int(rng.integers(1, len(c) + 1))]
    elif mode == 2:                                          # random bytes
        c = bytearray(rng.integers(0, 256, int(rng.integers(1, 400)), dtype=np.uint8).tobytes())
    else:                                                    # a valid stream with the wrong size announced
        pass
    payloads.append(bytes(c))
    sizes.append(len(x) if mode != 3 else max(0, len(x) + int(rng.integers(-40, 41))))
got, status = T.emul_inflate(payloads, sizes, None, guarded=True)
print("ran", len(payloads), "refused", int((status != 0).sum()))
"""


@pytest.mark.parametrize("seed", range(3))
def test_damaged_streams_stay_inside_their_buffers(seed):
    """Flipped, truncated, random and mis-sized streams with the compressed and the inflated buffer ending exactly
    gd::INF_SLACK bytes before an inaccessible page: the kernel must neither read nor write there (in a child process:
    a stray access is a segmentation fault), must terminate, and must not write behind the last member."""
    r = subprocess.run([sys.executable, "-c", GUARDED_FUZZ % H.ROOT, str(seed)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-2000:])
    assert "ran 192" in r.stdout and int(r.stdout.split("refused")[1]) > 100


def _libdeflate():
    try:
        ld = C.CDLL("libdeflate.so.0")
    except OSError:
        return None
    ld.libdeflate_alloc_compressor.restype = C.c_void_p
    ld.libdeflate_alloc_compressor.argtypes = [C.c_int]
    ld.libdeflate_deflate_compress.restype = C.c_size_t
    ld.libdeflate_deflate_compress.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
    ld.libdeflate_free_compressor.argtypes = [C.c_void_p]
    return ld


@both_kernels
@pytest.mark.skipif(_libdeflate() is None, reason="no libdeflate.so.0 on this system")
def test_streams_libdeflate_writes():
    """tools/synth_bam.cpp deflates with libdeflate when the system has it (as htslib does when built against it), and
    real BAM files mostly come from it: its block splitting, its choice between stored / fixed / dynamic blocks and its
    match statistics are not zlib's.  Streams of every level 1..12 over BAM-like records, text, runs and noise."""
    ld = _libdeflate()
    rng = np.random.default_rng(7)
    rec = bytearray()
    for i in range(260):                                       # records shaped like the synthetic BAM's
        rec += (295).to_bytes(4, "little") + bytes(8) + bytes([13, 60]) + rng.integers(0, 255, 6, dtype=np.uint8).tobytes()
        rec += (150).to_bytes(4, "little") + b"\xff" * 8 + bytes(4) + b"synth.%07d\0" % i + (150 << 4).to_bytes(4, "little")
        rec += rng.choice(np.frombuffer(bytes([0x11, 0x21, 0x41, 0x81, 0x12, 0x22, 0x42, 0x82]), np.uint8), 75).tobytes()
        rec += bytes(np.repeat(np.where(rng.integers(0, 8, 19) == 0, rng.integers(2, 37, 19), 37).astype(np.uint8), 8)[:150])
    rec = bytes(rec)
    parts, payloads = [], []
    texts = [rec[:0xff00], rec[:1000], b"", b"A", bytes(5000), rng.integers(0, 256, 3000, dtype=np.uint8).tobytes(),
             (b"the quick brown fox " * 400)[:7001], rng.integers(0, 4, 20000, dtype=np.uint8).tobytes()]
    for level in range(1, 13):
        co = ld.libdeflate_alloc_compressor(level)
        assert co
        for x in texts:
            buf = C.create_string_buffer(len(x) + len(x) // 8 + 256)
            n = ld.libdeflate_deflate_compress(co, x, len(x), buf, len(buf))
            assert n > 0
            c = buf.raw[:n]
            assert zlib.decompress(c, -15) == x
            parts.append(x); payloads.append(c)
        ld.libdeflate_free_compressor(co)
    got, status = emul_inflate(payloads, [len(x) for x in parts], [zlib.crc32(x) & 0xffffffff for x in parts], guarded=True)
    for i, x in enumerate(parts):
        assert status[i] == 0 and got[i] == x, (i, int(status[i]), len(x))


@both_kernels
def test_members_of_the_synthetic_bam(tmp_path, _kernel):
    """The file bench.py's `bam_file_scope` reads (tools/synth_bam.cpp -> goleft_amd/synth-bam), member by member: what the
    emulated kernel makes of each payload is what zlib makes of it, and the CRC the kernel checks is the trailer's."""
    import json
    import struct
    exe = os.path.join(H.ROOT, "goleft_amd", "synth-bam")
    if not os.path.exists(exe):
        pytest.skip("goleft_amd/synth-bam is not built")
    bam = str(tmp_path / "s.bam")
    info = json.loads(subprocess.check_output([exe, bam, "chrS", "150000,90000", "30", "20", "3"]).decode())
    raw = open(bam, "rb").read()
    assert info["bam_bytes"] == len(raw)
    parts, payloads, crcs, off = [], [], [], 0
    while off < len(raw):
        assert raw[off:off + 4] == b"\x1f\x8b\x08\x04" and raw[off + 12:off + 16] == b"BC\x02\x00"
        bsize, = struct.unpack_from("<H", raw, off + 16)
        c = raw[off + 18:off + bsize + 1 - 8]
        crc, isize = struct.unpack_from("<II", raw, off + bsize + 1 - 8)
        x = zlib.decompress(c, -15)
        assert len(x) == isize and zlib.crc32(x) & 0xffffffff == crc
        parts.append(x); payloads.append(c); crcs.append(crc)
        off += bsize + 1
    assert len(parts) > 150 and parts[-1] == b""                # the EOF member too
    if _kernel == 1:
        # (256 fibers per member: every sixth member, the first ones and the last)
        keep = sorted(set(range(0, len(parts), 6)) | {1, 2, len(parts) - 2, len(parts) - 1})
        parts, payloads, crcs = [parts[i] for i in keep], [payloads[i] for i in keep], [crcs[i] for i in keep]
    got, status = emul_inflate(payloads, [len(x) for x in parts], crcs, guarded=True)
    if _kernel == 1:
        assert fallbacks() <= 2                                 # (the empty EOF member is a fixed block of nothing: either kernel's)
    for i, x in enumerate(parts):
        assert status[i] == 0 and got[i] == x, (i, int(status[i]), len(x), info.get("deflate"))


def test_literal_pairs_at_every_length_and_alignment():
    """The loop emits two literals per iteration when the symbol behind a literal is a literal too.  Streams of nothing but
    literals (Z_HUFFMAN_ONLY) of every length 0..200 -- a pair that would end one byte past the member must not be taken,
    the pending bytes reach fifteen and sixteen in every phase against the 64-byte blocks (the members lie back to back:
    every start alignment occurs) -- and literals between short matches (a pair is not taken on top of a chunk)."""
    rng = np.random.default_rng(11)
    parts, how = [], []
    for n in range(0, 201):
        parts.append(rng.integers(0, 256, n, dtype=np.uint8).tobytes()); how.append((1, zlib.Z_HUFFMAN_ONLY))
    for n in (4095, 4096, 4097, 65280):
        parts.append(rng.integers(0, 7, n, dtype=np.uint8).tobytes()); how.append((6, zlib.Z_HUFFMAN_ONLY))
    for k in range(1, 40):                                    # k literals, a match of 3..20 bytes, k literals, ...
        unit = rng.integers(0, 256, 20, dtype=np.uint8).tobytes()
        x = b"".join(rng.integers(0, 256, k, dtype=np.uint8).tobytes() + unit[:3 + (i % 18)] for i in range(60))
        parts.append(unit + x); how.append((6, zlib.Z_DEFAULT_STRATEGY))
    check(parts, how)


def test_crc_kernels_at_slice_boundaries():
    """The wave-per-member CRC kernel cuts a member into 1 KB slices aligned to its END: lengths around every kind of
    boundary (one slice, a first slice of 1 / 1023 / 1024 bytes, BGZF's 65 280 and the format's 65 536), a wrong trailer
    in each (both kernels must say 18; emul_inflate fails when they disagree)."""
    rng = np.random.default_rng(5)
    parts, how = [], []
    for n in (1, 7, 8, 9, 1023, 1024, 1025, 2047, 2048, 2049, 3071, 4096, 32767, 32768, 65279, 65280, 65281, 65535, 65536):
        parts.append(rng.integers(0, 256, n, dtype=np.uint8).tobytes()); how.append((0 if n > 40000 else 1, zlib.Z_DEFAULT_STRATEGY))
    check(parts, how)
    payloads = [deflate(x, lv, st) for x, (lv, st) in zip(parts, how)]
    for flip in (0, 1):
        crcs = [(zlib.crc32(x) ^ (1 << (i % 32) if (i + flip) % 2 else 0)) & 0xffffffff for i, x in enumerate(parts)]
        got, status = emul_inflate(payloads, [len(x) for x in parts], crcs)
        for i, x in enumerate(parts):
            assert int(status[i]) == (18 if (i + flip) % 2 else 0) and got[i] == x, (i, len(x), int(status[i]))


def _stored_then_compressed(stored: bytes, tail: bytes, level, strategy) -> bytes:
    """One deflate stream: a non-final STORED block holding `stored`, then `tail` deflated with `stored` as its history
    (a preset dictionary in a raw stream is exactly that) -- what zlib and libdeflate emit when an incompressible stretch
    is followed by data that repeats it: no history reset between the blocks."""
    assert 0 < len(stored) <= 0xffff
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy, zdict=stored)
    body = co.compress(tail) + co.flush()
    return b"\x00" + len(stored).to_bytes(2, "little") + (len(stored) ^ 0xffff).to_bytes(2, "little") + stored + body


@both_kernels
def test_matches_that_reach_back_into_a_stored_block():
    """after a stored block the output ring held only the bytes of the block's last, incomplete
    64-byte block; a match whose 16-byte source straddles that boundary took the ring path and read bytes the ring never
    got.  Stored block + fixed / dynamic block whose first match lies 16..96 bytes back, at every alignment of the
    member in memory (the members lie back to back) and every phase of the stored block's end against the 64-byte blocks."""
    rng = np.random.default_rng(23)
    payloads, parts = [], []
    for d in range(16, 97):
        for rep in range(3):
            ns = int(rng.integers(d, 400))                     # the stored block: incompressible
            stored = rng.integers(0, 256, ns, dtype=np.uint8).tobytes()
            ml = int(rng.integers(3, 60))
            src = stored[ns - d:][:ml]
            tail = (src * (ml // len(src) + 1))[:ml] + rng.integers(0, 256, int(rng.integers(0, 30)), dtype=np.uint8).tobytes()
            tail += stored[-int(rng.integers(16, min(ns, 96) + 1)):][:40]
            st = (zlib.Z_FIXED, zlib.Z_DEFAULT_STRATEGY)[rep % 2]
            c = _stored_then_compressed(stored, tail, 6, st)
            x = stored + tail
            d0 = zlib.decompressobj(-15)
            assert d0.decompress(c) + d0.flush() == x and d0.eof
            payloads.append(c); parts.append(x)
    # ... and a pad member of every length 0..63 in front of the same stream: every obase
    stored = rng.integers(0, 256, 100, dtype=np.uint8).tobytes()
    for pad in range(64):
        for d in (16, 17, 31, 33, 63, 70):
            tail = stored[100 - d:][:24] + b"xyz"
            payloads.append(deflate(bytes(pad), 1)); parts.append(bytes(pad))
            payloads.append(_stored_then_compressed(stored, tail, 6, zlib.Z_FIXED)); parts.append(stored + tail)
    got, status = emul_inflate(payloads, [len(x) for x in parts], [zlib.crc32(x) & 0xffffffff for x in parts], guarded=True)
    bad = [(i, int(status[i])) for i, x in enumerate(parts) if status[i] != 0 or got[i] != x]
    assert not bad, (len(bad), len(parts), bad[:10])


def _bamlike(rng, n):
    """records the way a BAM holds them: a fixed-shape head, a name that differs from the previous one in a few digits, packed
    bases (random), qualities in plateaus -- matches at a record's distance, runs, and literals"""
    out, k = [], 0
    while sum(len(x) for x in out) < n:
        k += int(rng.integers(1, 400))
        head = np.array([0, 0, 0, 0], np.uint8).tobytes() + int(k).to_bytes(4, "little") + bytes([15, 60, 73, 18, 1, 0, 99, 0, 151, 0, 0, 0])
        name = b"A00741:188:HGTMNDSX2:3:%04d:%05d:%05d\0" % (int(rng.integers(1101, 2678)), int(rng.integers(1000, 33000)), int(rng.integers(1000, 37000)))
        seq = (rng.integers(0, 4, 76, dtype=np.uint8) * 0 + (1 << rng.integers(0, 4, 76)).astype(np.uint8) * 17 % 255).astype(np.uint8).tobytes()
        qual = b"".join(bytes([int(rng.integers(2, 41))]) * int(rng.integers(1, 60)) for _ in range(6))[:151].ljust(151, b"%")
        out.append(head + name + seq + qual)
    return b"".join(out)[:n]


def test_the_workgroup_per_member_kernel_takes_real_members_itself():
    """What GD_OPT_INFLATE_KERNEL = 1 is for: full-size members of BAM-like records -- one Huffman block (libdeflate's way), several
    (zlib at level 6 on 64 KB), with a stored block in between -- are inflated by the workgroup-per-member kernel ITSELF (no
    fallback), byte for byte zlib's, next to the lane-per-member kernel on the same streams."""
    rng = np.random.default_rng(11)
    parts = [_bamlike(rng, 65280) for _ in range(4)] + [_bamlike(rng, int(rng.integers(2000, 60000))) for _ in range(4)]
    pay = [deflate(x, lv) for x, lv in zip(parts, (1, 6, 9, 4, 1, 6, 9, 2))]
    # several blocks in one member: a sync flush every 9 KB (dynamic blocks with history across them), one of them stored
    x = _bamlike(rng, 65000)
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    multi = b"".join(co.compress(x[i:i + 9000]) + co.flush(zlib.Z_SYNC_FLUSH) for i in range(0, 54000, 9000))
    co2 = co.copy()
    multi += co.compress(x[54000:]) + co.flush()
    parts.append(x); pay.append(multi)
    y = rng.integers(0, 256, 3000, dtype=np.uint8).tobytes()
    parts.append(y + x[:30000]); pay.append(_stored_then_compressed(y, x[:30000], 6, zlib.Z_DEFAULT_STRATEGY))
    crcs = [zlib.crc32(x) & 0xffffffff for x in parts]
    res = {}
    for k in (0, 1):
        _lib().emul_inflate_kernel(k)
        got, status = emul_inflate(pay, [len(x) for x in parts], crcs, guarded=True)
        assert (status == 0).all() and got == parts, (k, status)
        res[k] = fallbacks()
    assert res[1] == 0, "the workgroup-per-member kernel left %d of %d members to the other kernel" % (res[1], len(parts))


def test_the_workgroup_per_member_kernel_hands_over_what_it_does_not_take():
    """Members it must not judge go to the lane-per-member kernel with WV_FALLBACK and come back with THAT kernel's status: a
    payload cut short, a distance that reaches in front of the member, an over-subscribed code, an incompressible member (its
    payload does not fit beside the output it has to produce)."""
    rng = np.random.default_rng(12)
    x = _bamlike(rng, 40000)
    good = deflate(x, 6)
    noise = rng.integers(0, 256, 65280, dtype=np.uint8).tobytes()
    cases = [(good[:len(good) // 2], len(x)),                # truncated
             (deflate(b"abcdefgh" * 100, 6)[:-4] + b"\xff\xff\xff\xff", 800),
             (good, len(x) - 7),                              # inflates to more than the member says
             (deflate(noise, 1), len(noise))]                 # stored / barely compressed: handled by either kernel
    _lib().emul_inflate_kernel(0)
    want_got, want_status = emul_inflate([c for c, _ in cases], [n for _, n in cases], None, guarded=True)
    _lib().emul_inflate_kernel(1)
    got, status = emul_inflate([c for c, _ in cases], [n for _, n in cases], None, guarded=True)
    assert list(status) == list(want_status), (list(status), list(want_status))
    assert status[3] == 0 and got[3] == noise
    for i in range(len(cases)):
        if want_status[i] == 0:
            assert got[i] == want_got[i]
