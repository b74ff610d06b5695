"""CPU, world_size 2 over gloo: chromosome sharding + the final gather.

The per-rank results are produced by the oracle here (test infrastructure);
on the GPU box the same gather runs on engine-owned device buffers over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from goleft_amd import shard
from oracle import pyoracle as po
from tests import helpers as H


def test_lpt_assign_hg19_matches_survey():
    from goleft_amd import synth
    L = synth.HG19_LENGTHS
    for n, bound in ((8, 401853479), (4, None), (2, None), (1, None)):
        a = shard.lpt_assign(L, n)
        assert sorted(t for r in a for t in r) == list(range(24))
        loads = [sum(L[t] for t in r) for r in a]
        if bound:
            assert max(loads) == bound          # max shard 401 853 479 bp
        assert max(loads) / (sum(L) / n) < 1.04
    assert shard.lpt_assign(L, 8) == shard.lpt_assign(L, 8)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_case():
    rng = np.random.default_rng(11)
    lengths = [9000, 4000, 12000, 700, 5000]
    reads = {t: H.random_reads(rng, l, 600) for t, l in enumerate(lengths) if t != 3}
    return lengths, reads


def _local(lengths, reads, tids, W, mincov, step):
    sums, mins, bounds = [], [], []
    for j, t in enumerate(tids):
        d = po.perbase_c(reads.get(t, H.empty_reads()), 1, 0, lengths[t])
        s, m = H.oracle_windows(d, W)
        r = H.oracle_runs(d, mincov, 0, step)
        sums.append(s)
        mins.append(m)
        bounds.append(np.stack([r[:, 0], r[:, 2] | (j << 2)], 1).astype(np.int32))
    cat = lambda xs, dt: np.concatenate(xs).astype(dt) if xs else np.zeros(0, dt)
    b = np.concatenate(bounds) if bounds else np.zeros((0, 2), np.int32)
    return cat(sums, np.int64), cat(mins, np.int32), b.reshape(-1)


def _worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lengths, reads = _make_case()
        W, mincov, step = 250, 4, 2500
        assignment = shard.lpt_assign(lengths, world)
        s, m, b = _local(lengths, reads, assignment[rank], W, mincov, step)
        g = shard.gather_to_root(torch.from_numpy(s), torch.from_numpy(m), torch.from_numpy(b),
                                 assignment, lengths, W, rank, world)
        if rank == 0:
            res = shard.unpack_gathered(g)
            ok = sorted(res) == list(range(len(lengths)))
            for t, l in enumerate(lengths):
                d = po.perbase_c(reads.get(t, H.empty_reads()), 1, 0, l)
                ws, wm = H.oracle_windows(d, W)
                wr = H.oracle_runs(d, mincov, 0, step)
                ok &= np.array_equal(res[t]["sums"].numpy(), ws)
                ok &= np.array_equal(res[t]["mins"].numpy(), wm)
                ok &= np.array_equal(res[t]["bounds"].numpy(), wr[:, [0, 2]])
            open(out_path, "w").write("ok" if ok else "mismatch")
        else:
            assert g is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gather_world(world, tmp_path):
    out = str(tmp_path / "res.txt")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert open(out).read() == "ok"


def test_gather_single_rank_is_identity():
    lengths, reads = _make_case()
    assignment = shard.lpt_assign(lengths, 1)
    s, m, b = _local(lengths, reads, assignment[0], 100, 3, 1000)
    g = shard.gather_to_root(torch.from_numpy(s), torch.from_numpy(m), torch.from_numpy(b),
                             assignment, lengths, 100, 0, 1)
    res = shard.unpack_gathered(g)
    off = 0
    for t, l in enumerate(lengths):
        nw = shard.n_windows(l, 100)
        assert np.array_equal(res[t]["sums"].numpy(), s[off:off + nw])
        off += nw


def _worker_steps(rank, world, port, out_path):
    """RootGather reused over several steps with CHANGING results (two fixed buffer pairs used alternately,
    one asynchronous collective per step), then a step whose boundary count exceeds the agreed capacity."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lengths, reads = _make_case()
        W, step = 250, 2500
        assignment = shard.lpt_assign(lengths, world)
        g = shard.RootGather(assignment, lengths, W, rank, world, torch.device("cpu"), bounds_cap=0)
        s, m, b = _local(lengths, reads, assignment[rank], W, 4, step)
        cap = g.reserve(len(b) // 2)
        send_ptrs = {b.data_ptr() for b in g._sends}   # two fixed send buffers, used alternately
        ok = True
        for mincov in (4, 2, 7, 4):                    # different class runs every step, same buffers
            s, m, b = _local(lengths, reads, assignment[rank], W, mincov, step)
            g.step(torch.from_numpy(s), torch.from_numpy(m), torch.from_numpy(b))
            ok &= g.send.data_ptr() in send_ptrs and {b.data_ptr() for b in g._sends} == send_ptrs   # never re-allocated
            if rank == 0:
                gg = g.result()
                ok &= not gg["overflow"]
                res = shard.unpack_gathered(gg)
                for t, l in enumerate(lengths):
                    d = po.perbase_c(reads.get(t, H.empty_reads()), 1, 0, l)
                    ws, wm = H.oracle_windows(d, W)
                    wr = H.oracle_runs(d, mincov, 0, step)
                    ok &= np.array_equal(res[t]["sums"].numpy(), ws) and np.array_equal(res[t]["mins"].numpy(), wm)
                    ok &= np.array_equal(res[t]["bounds"].numpy(), wr[:, [0, 2]])
        # more boundaries than the capacity: reported, never silently truncated
        big = np.zeros(2 * (cap + 5), np.int32)
        g.step(torch.from_numpy(s), torch.from_numpy(m), torch.from_numpy(big) if rank == world - 1 else torch.from_numpy(b))
        if rank == 0:
            gg = g.result()
            ok &= gg["overflow"] and gg["true_counts"][world - 1] == cap + 5
            open(out_path, "w").write("ok" if ok else "mismatch")
    finally:
        dist.destroy_process_group()


def test_root_gather_reused_over_steps(tmp_path):
    out = str(tmp_path / "res.txt")
    mp.spawn(_worker_steps, args=(2, _free_port(), out), nprocs=2, join=True)
    assert open(out).read() == "ok"
