"""-m gpu: the library's own collective (gd_comm_init / gd_gather_export, goleft_amd/csrc/gd_api_comm.inc) with a world of
TWO and THREE ranks -- the counterpart of the reference's merge loop (/root/reference/depth/depth.go:394-421) inside the C ABI.

The development boxes have one GPU and RCCL refuses two ranks on one device, so until the driver's 8-GPU run these calls had
only ever had a world of one (tests/test_gpu_export.py).  Here every rank is a process of its own with an engine context on
device 0, and the eight RCCL entry points the library binds are served by tests/stubs/rccl_stub.cpp -- sends and receives
between processes over POSIX shared memory -- named through GOLEFT_RCCL_LIB.  Checked: the id's 128 bytes passed by value,
the offsets of the root's grouped receives (rank r's block at r * words), a different result in every step with two send
buffers alternating, LPT shards of different sizes in one fixed-size block."""
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu

STUB_DIR = os.path.join(H.ROOT, "tests", "stubs")


def stub():
    so, src = os.path.join(STUB_DIR, "librccl_stub.so"), os.path.join(STUB_DIR, "rccl_stub.cpp")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["hipcc", "-O2", "-shared", "-fPIC", "-o", so, src, "-lrt"])
    return so


@pytest.mark.parametrize("world", [2, 3])
def test_gather_export_between_processes(tmp_path, world):
    steps = 5
    env = dict(os.environ, GOLEFT_RCCL_LIB=stub())
    procs = [subprocess.Popen([sys.executable, os.path.join(H.ROOT, "tests", "gather_world_child.py"), str(tmp_path), str(r), str(world), str(steps)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=600))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d: %s\n%s" % (r, so[-1000:], se[-3000:])
    seen = set()
    for s in range(steps):
        root = np.load(os.path.join(str(tmp_path), "root_%d.npy" % s))
        words = len(root) // world
        for r in range(world):
            own = np.load(os.path.join(str(tmp_path), "own_%d_%d.npy" % (r, s)))
            assert len(own) == words
            assert np.array_equal(root[r * words:(r + 1) * words], own), "step %d: rank %d's block is not at %d * words of the root's buffer" % (s, r, r)
            assert own[0] > 0                                     # class boundaries were exported
            seen.add((r, hash(own[:4096].tobytes())))
    # the steps really differed (a gather that kept delivering step 0's buffer would have passed the comparison above otherwise)
    assert len(seen) >= world * 3
