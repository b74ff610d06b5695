"""CPU: the shape of the inflate kernel's machine code.  The kernel is fast because its memory waits are COUNTED
(goleft_amd/csrc/gd_inflate.hpp): the chunk load and the input word are issued at the top of an iteration (round 5: above
the block-header path), one symbol is decoded while they are in flight, and the first wait for vector memory after them
in the symbol loop is the explicit `s_waitcnt vmcnt(0)` behind the decode.  That
property is the compiler's to give and to take: a harmless-looking edit (moving the block-header code into a function
was tried) changed the register allocation and put an `s_waitcnt vmcnt(0)` into the path that builds a chunk from the
register window -- a full store round trip in every iteration that starts a short-distance match -- without failing any
functional test.  This test compiles the kernel for gfx950 and checks the schedule, so such a change is seen on the
CPU box."""
import os
import re
import shutil
import subprocess

import pytest

from tests import helpers as H

HIPCC = next((p for p in ("/opt/rocm/bin/hipcc", shutil.which("hipcc") or "") if p and os.path.exists(p)), None)

pytestmark = pytest.mark.skipif(HIPCC is None, reason="needs hipcc")


def test_waits_of_the_symbol_loop_are_counted(tmp_path):
    src = tmp_path / "k.hip"
    src.write_text('#include <hip/hip_runtime.h>\n#include <cstdint>\n#include "%s"\n' %
                   os.path.join(H.ROOT, "goleft_amd", "csrc", "gd_inflate.hpp"))
    asm = tmp_path / "k.s"
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", str(asm), str(src)],
                          stderr=subprocess.DEVNULL)
    text = asm.read_text()
    body = text[text.index("gd_inflate_kernel"):text.index("s_endpgm")].splitlines()
    # the chunk load (the kernel's only 16-byte load with the streaming hint) and the input word right after it
    loads = [i for i, l in enumerate(body) if "global_load_dwordx4" in l and " nt" in l]
    assert len(loads) == 1, loads
    at = loads[0]
    # round 4: the input comes through a 64-byte window in LDS; its refill -- one 16-byte load, by the lanes whose oldest
    # slot is consumed -- is issued behind the chunk load (and the ring reads of the lanes whose match source has not been
    # stored yet), and the bit buffer's refill reads LDS
    assert any("global_load_dwordx4" in l and " nt" not in l for l in body[at + 1:at + 120]), "the window's refill is not issued with the chunk load"
    stores = [i for i, l in enumerate(body) if "global_store_dwordx4" in l and i > at]
    assert stores, "no 16-byte store after the loads"
    # round 5: the block-header path lies BETWEEN the two loads and the decode; it ends with the restart of the input window
    # (four 16-byte loads), behind which the symbol loop proper begins
    hdr_end = max(i for i, l in enumerate(body[:stores[0]]) if "global_load_dwordx4" in l and " nt" not in l)
    assert at + 200 < hdr_end < stores[0] and sum("global_load_dwordx4" in l for l in body[hdr_end - 8:hdr_end + 1]) == 4
    hdr_end = next(i for i in range(hdr_end, stores[0]) if "s_waitcnt vmcnt(0)" in body[i]) + 1   # (... and their four window writes)
    assert not any("global_load_dwordx2" in l for l in body[hdr_end:stores[0]]), "an 8-byte input load is back in the symbol loop"
    # the output leaves in whole 64-byte blocks: four 16-byte stores to one block, back to back
    assert len(stores) >= 4 and stores[3] - stores[0] <= 8, stores[:6]
    offs = [re.search(r"offset:(\d+)", body[i]) for i in stores[:4]]
    assert [int(m.group(1)) if m else 0 for m in offs] == [0, 16, 32, 48], [body[i].strip() for i in stores[:4]]
    # no wait for vector memory between the loop's head and the loads: a load that some path leaves pending (a use inside a
    # branch) costs an `s_waitcnt vmcnt(0)` in front of the next iteration's loads, behind this one's block stores
    head = max(i for i, l in enumerate(body[:at]) if "Loop Header" in l)
    assert not any("vmcnt" in l for l in body[head:at]), [l.strip() for l in body[head:at] if "vmcnt" in l]
    between = body[hdr_end:stores[0]]
    waits = [l.strip() for l in between if "s_waitcnt" in l and "vmcnt" in l]
    # both loads are conditional (few lanes refill in an iteration; a source may come from the ring), so the waits for them
    # are vmcnt(0) -- the first after a whole symbol has been decoded (the explicit one), the others where the compiler
    # cannot see that it has happened (the chunk's append, the window slot's write)
    assert waits and set(waits) == {"s_waitcnt vmcnt(0)"} and len(waits) <= 3, waits
    # the decode between the loads and that first wait is long: a whole symbol (both Huffman look-ups)
    first_wait = next(i for i, l in enumerate(between) if "vmcnt" in l)
    assert first_wait > 120, first_wait
    assert sum("ds_read" in l for l in between[:first_wait]) >= 4
    # the chunk built from the register window (two 16-byte selector reads, byte permutes) waits for LDS only
    perm = [i for i, l in enumerate(body) if "ds_read_b128" in l]
    assert len(perm) == 2 and perm[1] == perm[0] + 1
    window = body[perm[0]:perm[0] + 30]
    assert sum("v_perm_b32" in l for l in window) == 8
    assert not any("vmcnt" in l for l in window), [l.strip() for l in window if "vmcnt" in l]
    # occupancy is set by LDS (six workgroups per CU); the registers must allow at least two waves per SIMD
    vgprs = int(re.search(r"\.vgpr_count:\s+(\d+)", text[text.index("gd_inflate_kernel"):]).group(1))
    assert vgprs <= 256, vgprs
    lds = int(re.search(r"\.group_segment_fixed_size:\s+(\d+)", text).group(1))
    assert lds * 4 <= 160 * 1024, lds
