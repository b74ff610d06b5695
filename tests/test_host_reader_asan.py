"""CPU: the host BAM reader under AddressSanitizer + UndefinedBehaviorSanitizer on damaged files (record stream, .bai,
BGZF bytes, truncation), with both inflate back ends and with batches small enough that every file spans several.  The
hypothesis tests in test_host_cpu.py call the same reader in the plain build, where a stray read only shows if it crashes."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from oracle import bamio
from tests import helpers as H

GXX = shutil.which("g++")
HOST = os.path.join(H.ROOT, "goleft_amd", "csrc", "host")


@pytest.fixture(scope="module")
def asan_reader(tmp_path_factory):
    if GXX is None:
        pytest.skip("no g++")
    exe = str(tmp_path_factory.mktemp("asan") / "reader_asan")
    r = subprocess.run([GXX, "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-std=c++17", "-pthread", "-I", HOST, "-o", exe,
                        os.path.join(H.ROOT, "tests", "emul", "reader_asan_main.cpp"), os.path.join(HOST, "bam_reader.cpp"),
                        "-lz", "-ldl"], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("this toolchain has no sanitizer run time: " + r.stderr[-300:])
    return exe


def test_reader_stays_inside_its_buffers(asan_reader, tmp_path):
    files = []
    for seed in range(40):
        rng = np.random.default_rng(1000 + seed)
        contigs = [("f1", 30_000), ("f2", 9_000)]
        reads = {0: H.random_reads(rng, 30_000, 300, max_len=80), 1: H.long_cigar_reads(rng, 9_000, [70_000, 3], max_step=2)}
        p = str(tmp_path / ("v%d.bam" % seed))
        bamio.write_bam(p, contigs, reads, unplaced=1, index=True, level=int(rng.integers(0, 7)))
        good = open(p, "rb").read()
        raw = bytearray(bamio.bgzf_decompress(good))
        hdr = 12 + int.from_bytes(raw[4:8], "little") + sum(8 + len(n) + 1 for n, _ in contigs)
        for _ in range(int(rng.integers(1, 12))):                   # the record stream (every third file: the header too)
            raw[int(rng.integers(hdr if seed % 3 else 0, len(raw)))] = int(rng.integers(0, 256))
        q = str(tmp_path / ("c%d.bam" % seed))
        open(q, "wb").write(bamio.bgzf_compress(bytes(raw)))
        bai = bytearray(open(p + ".bai", "rb").read())
        for _ in range(int(rng.integers(0, 6))):
            bai[int(rng.integers(4, len(bai)))] = int(rng.integers(0, 256))
        open(q + ".bai", "wb").write(bytes(bai))
        z = bytearray(good)                                             # the BGZF bytes themselves
        for _ in range(int(rng.integers(1, 6))):
            z[int(rng.integers(0, len(z)))] = int(rng.integers(0, 256))
        zp = str(tmp_path / ("z%d.bam" % seed))
        open(zp, "wb").write(bytes(z))
        tp = str(tmp_path / ("t%d.bam" % seed))                         # cut short
        open(tp, "wb").write(good[:int(rng.integers(1, len(good)))])
        files += [p, q, zp, tp]
    for env in ({"GOLEFT_BAM_CHUNK_KB": "64", "GOLEFT_BAM_HEAD_KB": "1"}, {}, {"GOLEFT_HOST_ZLIB": "1", "GOLEFT_BAM_CHUNK_KB": "64"}):
        r = subprocess.run([asan_reader] + files, capture_output=True, text=True, env=dict(os.environ, **env))
        assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (env, r.stderr[-3000:])
