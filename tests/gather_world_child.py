"""TEST INFRASTRUCTURE: one rank of tests/test_gpu_gather_world.py -- a process of its own with an engine context on device 0,
the library's collective bound to the RCCL stand-in (GOLEFT_RCCL_LIB = tests/stubs/librccl_stub.so).  No torch in here: what a
cgo host has is the C ABI alone (gd_device_alloc / gd_set_export / gd_comm_init / gd_gather_export / gd_device_read).

    python tests/gather_world_child.py <dir> <rank> <world> <steps>"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    d, rank, world, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    from goleft_amd import shard, synth
    from goleft_amd.engine import DepthEngine, comm_library, comm_unique_id
    lib_used = comm_library()
    assert lib_used and lib_used[0].endswith("librccl_stub.so") and not lib_used[1], lib_used
    lengths = [1_500_000, 700_000, 2_300_000, 400_000, 1_100_000, 900_000, 1_900_000]
    W = 1000
    mine = shard.lpt_assign(lengths, world)[rank]
    max_w = max(sum(shard.n_windows(lengths[t], W) for t in ts) for ts in shard.lpt_assign(lengths, world))
    cap_b = 1 << 17
    words = 1 + max_w + (max_w + 1) // 2 + cap_b
    with DepthEngine(0) as eng:
        lib = eng._lib
        eng.set_contigs([lengths[t] for t in mine])
        for k, t in enumerate(mine):
            eng.push(k, *synth.short_reads_numpy(lengths[t], synth.n_reads_for(lengths[t], 10.0), 100 + t))
        bufs = []
        for _ in range(2):                                   # two send buffers alternate, as gd_gather_export's contract says
            p = C.c_void_p()
            assert lib.gd_device_alloc(eng._ctx, 8 * words, C.byref(p)) == 0
            bufs.append(p)
        recv = C.c_void_p()
        if rank == 0:
            assert lib.gd_device_alloc(eng._ctx, 8 * words * world, C.byref(recv)) == 0
        # the 128 bytes travel by whatever means the host has: here a file
        idf = os.path.join(d, "uid.bin")
        if rank == 0:
            uid = comm_unique_id()
            with open(idf + ".tmp", "wb") as fh:
                fh.write(uid)
            os.rename(idf + ".tmp", idf)
        t0 = time.time()
        while not os.path.exists(idf):
            assert time.time() - t0 < 60, "rank 0 never wrote the id"
            time.sleep(0.01)
        uid = open(idf, "rb").read()
        eng.comm_init(rank, world, uid)
        for s in range(steps):
            # every step computes something else (the MAPQ threshold moves: 1 % of the reads have MAPQ 0, others 60), so a
            # stale buffer cannot pass for a fresh one
            eng.set_params(window_size=W, min_mapq=(1, 0, 61, 30)[s % 4], min_cov=4 + s)
            eng.set_export(bufs[s & 1].value, max_w, cap_b)
            eng.compute()
            eng.gather_export(recv_ptr=recv.value or 0, words=words, root=0)
            eng.gather_wait()
            own = np.empty(words, np.int64)
            assert lib.gd_device_read(eng._ctx, own.ctypes.data, bufs[s & 1], 8 * words) == 0
            np.save(os.path.join(d, "own_%d_%d.npy" % (rank, s)), own)
            if rank == 0:
                got = np.empty(words * world, np.int64)
                assert lib.gd_device_read(eng._ctx, got.ctypes.data, recv, 8 * words * world) == 0
                np.save(os.path.join(d, "root_%d.npy" % s), got)
        eng.set_export(0, 0, 0)
        eng.comm_destroy()
        for p in bufs + ([recv] if rank == 0 else []):
            assert lib.gd_device_free(eng._ctx, p) == 0
    print("rank %d of %d done: %d words per block" % (rank, world, words))


if __name__ == "__main__":
    main()
