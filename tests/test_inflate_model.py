"""CPU: tools/inflate_model.py -- what the lanes of the inflate kernel do per iteration on a BAM's members, counted by the
host emulation (tests/emul/inflate_stats.cpp: the kernel SOURCE with its probe macro switched on).  DESIGN.md 3.5 quotes its
numbers; this keeps the tool running against the kernel as it is."""
import json
import os
import shutil
import subprocess
import sys

import pytest

from tests import helpers as H

CLANG = next((p for p in ("/opt/rocm/lib/llvm/bin/clang++", shutil.which("clang++") or "") if p and os.path.exists(p)), None)

pytestmark = pytest.mark.skipif(CLANG is None, reason="the kernel source needs clang (ext_vector_type) to compile for the host")


def test_model_of_a_small_synthetic_bam(tmp_path):
    exe = os.path.join(H.ROOT, "goleft_amd", "synth-bam")
    if not os.path.exists(exe):
        pytest.skip("goleft_amd/synth-bam is not built")
    bam = str(tmp_path / "m.bam")
    subprocess.check_call([exe, bam, "chrS", "1500000", "30", "20"], stdout=subprocess.DEVNULL)
    out = subprocess.check_output([sys.executable, os.path.join(H.ROOT, "tools", "inflate_model.py"), bam, "--waves", "2", "--json"], timeout=600)
    d = json.loads(out.decode().strip().splitlines()[-1])
    assert d["waves"] == 2 and d["members"] == 128
    assert d["output_bytes"] == 128 * 65280 and 0 < d["input_bytes"] < d["output_bytes"]
    # a 64 KB member of this data is ~13 000 iterations of ~4-5 bytes each (two literals or a 16-byte chunk at most)
    assert 5000 < d["iterations_per_wave"]["mean"] < 40000 and 2.0 < d["output_bytes_per_lane_iteration"] <= 16.0
    li = d["lane_iterations"]
    assert abs(sum(li.values()) - 1.0) < 1e-6 and li["decode_a_symbol"] > 0.5
    # every chunk of a match has one source: memory or the ring
    assert d["chunks_from_memory_per_member"] > 0 and d["chunks_from_the_ring_per_member"] > 0
    cdf = d["chunk_loads_from_memory_with_distance_at_most"]
    vals = [cdf[k] for k in sorted(cdf, key=int)]
    assert vals == sorted(vals) and abs(vals[-1] - 1.0) < 1e-9 and vals[0] < 0.2   # (sources closer than 128 bytes come from the ring)


def test_huffman_self_synchronisation_tool(tmp_path):
    """tools/huffman_sync.py: its DEFLATE parser reproduces zlib's output length for every member it looks at (asserted inside),
    and a decoder started at a wrong bit offset falls into step within a few dozen symbols -- the figure DESIGN.md section 7
    builds the wave-per-member design on."""
    exe = os.path.join(H.ROOT, "goleft_amd", "synth-bam")
    if not os.path.exists(exe):
        pytest.skip("goleft_amd/synth-bam is not built")
    bam = str(tmp_path / "m.bam")
    subprocess.check_call([exe, bam, "chrS", "1500000", "30", "20"], stdout=subprocess.DEVNULL)
    out = subprocess.check_output([sys.executable, os.path.join(H.ROOT, "tools", "huffman_sync.py"), bam, "--members", "4", "--starts", "60",
                                   "--skip", "7"], timeout=600)
    d = json.loads(out.decode())
    assert d["members"] == 4 and d["huffman_blocks_per_member"] >= 1.0 and d["symbols_per_member"] > 5000
    s = d["symbols_until_in_step"]
    assert 1 <= s["median"] <= 30 and s["median"] <= s["p90"] <= s["p99"] <= s["max"] < 2000
    assert d["ran_into_an_invalid_code_or_the_block_end_first"] < 0.05
    q = d["sequential_symbol_steps_per_member"]
    assert q["64_lanes_speculative"] < q["one_lane"] / 10
