"""CPU: the device BAM record walk (goleft_amd/csrc/gd_bamdecode.hpp) compiled for the HOST and run lane by lane
(tests/emul/bamwalk_emul.cpp), the inflated stream and every output array ending exactly at an inaccessible page.  On
intact streams the two passes deliver what the records say; on damaged streams -- the fields a walk trusts flipped:
block_size, refID, POS, l_read_name, n_cigar_op, l_seq, the CIGAR, the CG tag -- the walk must terminate and never
touch a byte outside the stream or its outputs (a stray access is a segmentation fault of the child process).  The GPU
twin of this test (tests/test_gpu_bamdecode.py::test_device_bam_walker_mutation_fuzz) compares with the host decoder;
a read past the end is silent there."""
import ctypes as C
import os
import shutil
import struct
import subprocess
import sys

import numpy as np
import pytest

from oracle import bamio
from tests import helpers as H

EMUL_DIR = os.path.join(H.ROOT, "tests", "emul")
CLANG = next((p for p in ("/opt/rocm/lib/llvm/bin/clang++", shutil.which("clang++") or "") if p and os.path.exists(p)), None)

pytestmark = pytest.mark.skipif(CLANG is None, reason="the kernel source is compiled for the host with clang")


def _lib():
    src = os.path.join(EMUL_DIR, "bamwalk_emul.cpp")
    so = os.path.join(EMUL_DIR, "bamwalk_emul.so")
    deps = [src, os.path.join(EMUL_DIR, "emul_machine.hpp"), os.path.join(H.ROOT, "goleft_amd", "csrc", "gd_bamdecode.hpp")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call([CLANG, "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so, src])
    lib = C.CDLL(so)
    lib.emul_bam_walk.argtypes = ([C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int32, C.c_int32] + [C.c_void_p] * 7 +
                                  [C.c_uint64, C.c_uint64] + [C.c_void_p] * 5)
    lib.emul_bam_walk.restype = C.c_int
    lib.emul_bam_walk_mode(C.c_int(MODE))
    return lib


# how the records are extracted: 1 (what gd_api_ingest.inc does) the counting walk leaves a table of record starts and
# gd_bam_extract_tab_kernel takes a thread per record; 0 the second walk (kept for segments of 4 GB and more)
MODE = 1


@pytest.fixture(params=[1, 0], ids=["record-table", "second-walk"])
def both_extractions(request):
    global MODE
    MODE = request.param
    yield request.param
    MODE = 1


def record_starts(d: bytes):
    """(offset, refID) of every record of an inflated BAM."""
    l_text, = struct.unpack_from("<i", d, 4)
    p = 8 + l_text
    n_ref, = struct.unpack_from("<i", d, p)
    p += 4
    for _ in range(n_ref):
        l_name, = struct.unpack_from("<i", d, p)
        p += 8 + l_name
    out = []
    while p + 8 <= len(d):
        bs, ref = struct.unpack_from("<ii", d, p)
        out.append((p, ref))
        p += 4 + bs
    return out, n_ref


def walk(data: bytes, anchors, tid, n_ref):
    """The two passes over the segments [anchor_i, anchor_i+1) (the last one to the end of the stream).
    -> (rc, per-segment dict, arrays or None)"""
    lib = _lib()
    raw = np.frombuffer(data, np.uint8)
    beg = np.asarray(anchors, np.uint64)
    end = np.concatenate([beg[1:], [len(data)]]).astype(np.uint64)
    n = len(beg)
    n_rec = np.zeros(n, np.uint32); n_ops = np.zeros(n, np.uint64); first = np.zeros(n, np.int32); last = np.zeros(n, np.int32)
    flags = np.zeros(n, np.uint32)
    N = C.c_uint64(0); M = C.c_uint64(0)
    cap_rec, cap_ops = len(data) // 36 + 1, len(data) // 4 + 1
    pos = np.zeros(cap_rec, np.int32); flag = np.zeros(cap_rec, np.uint16); mapq = np.zeros(cap_rec, np.uint8)
    off = np.zeros(cap_rec + 1, np.uint32); cig = np.zeros(cap_ops, np.uint32)
    rc = lib.emul_bam_walk(raw.ctypes.data, len(data), beg.ctypes.data, end.ctypes.data, n, tid, n_ref, n_rec.ctypes.data,
                           n_ops.ctypes.data, first.ctypes.data, last.ctypes.data, flags.ctypes.data, C.byref(N), C.byref(M),
                           cap_rec, cap_ops, pos.ctypes.data, flag.ctypes.data, mapq.ctypes.data, off.ctypes.data, cig.ctypes.data)
    seg = dict(n_rec=n_rec, n_ops=n_ops, first=first, last=last, flags=flags)
    if rc != 0:
        return rc, seg, None
    off[N.value] = M.value
    return rc, seg, (pos[:N.value], flag[:N.value], mapq[:N.value], off[:N.value + 1], cig[:M.value])


def make_stream(tmp_path, seed, long_cigars=False):
    rng = np.random.default_rng(seed)
    contigs = [("c1", 300_000), ("c2", 120_000), ("c3", 50_000)]
    reads = {0: H.random_reads(rng, 300_000, 2500, max_len=150), 2: H.random_reads(rng, 50_000, 400, max_len=150)}
    if long_cigars:                                          # one CIGAR behind the CG:B,I tag (more than 65 535 ops)
        r = reads[0]
        ops = np.tile(np.asarray([(1 << 4) | 0, (1 << 4) | 2], np.uint32), 33_000)
        cig = np.concatenate([r.cigar, ops])
        off = np.concatenate([r.cigar_off, [r.cigar_off[-1] + len(ops)]]).astype(np.uint32)
        reads[0] = type(r)(np.concatenate([r.pos, [r.pos[-1]]]).astype(np.int32), np.concatenate([r.flag, [0]]).astype(np.uint16),
                           np.concatenate([r.mapq, [60]]).astype(np.uint8), off, cig)
    path = str(tmp_path / ("s%d.bam" % seed))
    bamio.write_bam(path, contigs, reads, unplaced=3)
    return bamio.bgzf_decompress(open(path, "rb").read()), contigs, reads


@pytest.mark.parametrize("long_cigars", [False, True])
def test_walk_delivers_the_records(tmp_path, long_cigars, both_extractions):
    data, contigs, reads = make_stream(tmp_path, 1, long_cigars)
    starts, n_ref = record_starts(data)
    for tid in (0, 2):
        mine = [p for p, ref in starts if ref == tid]
        for step in (1, 7, 64, 100_000):                     # an anchor per record ... one anchor for the whole contig
            rc, seg, arr = walk(data, mine[::step], tid, n_ref)
            assert rc == 0 and not (seg["flags"] & 7).any()
            r = reads[tid]
            assert int(seg["n_rec"].sum()) == r.n
            assert np.array_equal(arr[0], r.pos) and np.array_equal(arr[1], r.flag) and np.array_equal(arr[2], r.mapq)
            assert np.array_equal(arr[3], r.cigar_off) and np.array_equal(arr[4], r.cigar)
            # the contig's walk ends at the first record of another reference: bit 3 on the last segment, no resume
            assert seg["flags"][-1] & 8 and not (seg["flags"] & 16).any()
    # an anchor that is no record start: the walk overruns the next anchor (bit 2) or meets a corrupt record (bit 1)
    mine = [p for p, ref in starts if ref == 0]
    rc, seg, _ = walk(data, [mine[0], mine[5] + 3, mine[9]], 0, n_ref)
    assert rc == 1 and (seg["flags"] & 6).any()


FUZZ = r"""
import sys
import numpy as np
sys.path.insert(0, %r)
from tests import test_bamwalk_emul as T
import pathlib, tempfile
seed = int(sys.argv[1])
tmp = pathlib.Path(tempfile.mkdtemp())
data, contigs, reads = T.make_stream(tmp, 50 + seed, long_cigars=(seed %% 2 == 1))
starts, n_ref = T.record_starts(data)
mine = [p for p, ref in starts if ref == 0]
rng = np.random.default_rng(seed)
clean = refused = 0
for case in range(120):
    d = bytearray(data)
    for _ in range(int(rng.integers(1, 6))):
        s = mine[int(rng.integers(0, len(mine)))]
        # block_size, refID, POS, l_read_name, MAPQ / bin, n_cigar_op, flag, l_seq, the first CIGAR bytes, anywhere in the record
        at = s + [int(rng.integers(0, 4)), 4 + int(rng.integers(0, 4)), 8 + int(rng.integers(0, 4)), 12, 13 + int(rng.integers(0, 3)),
                  16 + int(rng.integers(0, 2)), 18 + int(rng.integers(0, 2)), 20 + int(rng.integers(0, 4)), 36 + int(rng.integers(0, 30)),
                  int(rng.integers(0, 300))][int(rng.integers(0, 10))]
        if at < len(d):
            d[at] = (d[at] ^ (1 << int(rng.integers(0, 8)))) if rng.random() < 0.6 else int(rng.integers(0, 256))
    if case %% 5 == 4:                                        # and the stream cut anywhere (a range that ends inside a record)
        d = d[:int(rng.integers(mine[0] + 1, len(d)))]
    anchors = [a for a in mine[::int(rng.integers(1, 40))] if a < len(d)] or [mine[0]]
    rc, seg, arr = T.walk(bytes(d), anchors, 0, n_ref)
    if rc == 0:
        clean += 1
        pos, flag, mapq, off, cig = arr
        assert len(off) == len(pos) + 1 and (np.diff(off.astype(np.int64)) >= 0).all() and off[-1] == len(cig)
    else:
        refused += 1
print("clean", clean, "refused", refused)
"""


@pytest.mark.parametrize("seed", range(4))
def test_damaged_streams_never_leave_their_buffers(seed):
    r = subprocess.run([sys.executable, "-c", FUZZ % H.ROOT, str(seed)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stdout[-300:], r.stderr[-2000:])
    clean, refused = int(r.stdout.split()[1]), int(r.stdout.split()[3])
    assert clean + refused == 120 and refused > 20
