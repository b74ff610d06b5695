"""CPU: the resources of the hot kernels as hipcc compiles them for gfx950 -- what decides how many waves a CU holds.
The tile kernels and the long-read kernel are tuned to seven 256-thread workgroups per CU (LDS) at no more than 64
vector registers, the streaming sums kernel and the inflate kernel to six; none of them may spill.  A change that costs
a register too many or a kilobyte of LDS passes every functional test and loses a seventh of the memory-level
parallelism; here it fails on the CPU box."""
import os
import re
import shutil
import subprocess

import pytest

from tests import helpers as H

HIPCC = next((p for p in ("/opt/rocm/bin/hipcc", shutil.which("hipcc") or "") if p and os.path.exists(p)), None)
LDS_PER_CU = 160 * 1024

pytestmark = pytest.mark.skipif(HIPCC is None, reason="needs hipcc")


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    asm = tmp_path_factory.mktemp("isa") / "api.s"
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                           "-I", os.path.join(H.ROOT, "include"), "-o", str(asm),
                           os.path.join(H.ROOT, "goleft_amd", "csrc", "gd_api.hip")], stderr=subprocess.DEVNULL)
    text = asm.read_text()
    meta = text[text.index("amdhsa.kernels:"):]
    names, out = [], {}
    for k in re.split(r"\n  - \.", meta)[1:]:
        g = lambda key: re.search(r"\.%s:\s+(\S+)" % key, "." + k).group(1)
        names.append(g("name"))
        out[names[-1]] = dict(lds=int(g("group_segment_fixed_size")), scratch=int(g("private_segment_fixed_size")),
                              sgpr=int(g("sgpr_count")), vgpr=int(g("vgpr_count")))
    dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.strip().splitlines()
    return {d: out[n] for d, n in zip(dem, names)}


def pick(kernels, prefix):
    got = {k: v for k, v in kernels.items() if prefix in k}
    assert got, prefix
    return got


def test_tile_kernels_keep_seven_workgroups_per_cu(kernels):
    for name, r in {**pick(kernels, "gd_tile_fast_kernel<"), **pick(kernels, "gd_ltile2_kernel<")}.items():
        assert r["scratch"] == 0, (name, r)
        assert r["lds"] * 7 <= LDS_PER_CU, (name, r)
        assert r["vgpr"] <= 64, (name, r)                    # 7 x 4 waves = 28 per CU = 7 per SIMD: 512 / 7 = 73 registers at most


def test_streaming_and_ingest_kernels(kernels):
    for name, r in pick(kernels, "gd_sums_stream_kernel").items():
        assert r["scratch"] == 0 and r["lds"] * 6 <= LDS_PER_CU and r["vgpr"] <= 84, (name, r)     # six waves per SIMD
    for name, r in {**pick(kernels, "gd_dels_raw_kernel"), **pick(kernels, "gd_prep_kernel<"), **pick(kernels, "gd_tile_slow_kernel<"),
                    **pick(kernels, "gd_h2d_kernel"), **pick(kernels, "gd_bam_walk_kernel<"),
                    **pick(kernels, "gd_index_records_kernel"), **pick(kernels, "gd_readback_kernel")}.items():
        assert r["scratch"] == 0, (name, r)


def test_inflate_kernel_keeps_four_workgroups_per_cu(kernels):
    # round 4: the lanes' input windows (and output rings) live in LDS next to the tables -- four to five workgroups per
    # CU, measured faster than the six the tables alone allowed (profiles/r10k_inflate_occupancy.txt, r10v_…)
    (name, r), = pick(kernels, "gd_inflate_kernel").items()
    assert r["lds"] * 4 <= LDS_PER_CU and r["vgpr"] <= 256, (name, r)      # LDS decides; two waves per SIMD at most
    assert r["scratch"] <= 512, (name, r)                    # the code lengths of a dynamic block (a lane's private array)


def test_workgroup_per_member_inflate_keeps_two_members_per_cu(kernels):
    """gd_inflate_wave_kernel<4>: a member's 64 KB of output, its tables, the lanes' words and the piece bitmap in LDS -- two
    workgroups of four waves per CU, so at most half of the LDS and 256 registers a lane (two waves per SIMD), and no array in
    scratch (an indexed one in pass B2 once cost it a memory round trip per piece: profiles/r13f_inflate_two_kernels.txt; a few
    spilled words between the phases are what its 300 scalar values cost at 256 registers)."""
    (name, r), = pick(kernels, "gd_inflate_wave_kernel<").items()
    assert r["lds"] * 2 <= LDS_PER_CU and r["vgpr"] <= 256 and r["scratch"] <= 32, (name, r)
