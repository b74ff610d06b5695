"""-m gpu: gd_set_export -- the packed result block a merge rank gathers (depth/depth.go:394-421)
is written by gd_compute itself and equals the separately fetched results, bit for bit."""
import numpy as np
import pytest

from oracle import pyoracle as po
from tests import helpers as H

pytestmark = pytest.mark.gpu


def test_export_block_equals_results():
    import torch
    from goleft_amd import shard
    from goleft_amd.engine import DepthEngine, GdError
    rng = np.random.default_rng(17)
    lengths = [50_000, 8_191, 120_001, 1, 33_000]
    reads = {t: H.random_reads(rng, l, 4000) for t, l in enumerate(lengths) if t != 3}
    W = 250
    nw = [shard.n_windows(l, W) for l in lengths]
    max_w = sum(nw) + 7
    words_m = (max_w + 1) // 2
    dev = torch.device("cuda", 0)
    for cap_b in (1 << 16, 5):                                   # roomy, then too small for the boundaries
        buf = torch.full((1 + max_w + words_m + cap_b,), -1, dtype=torch.int64, device=dev)
        with DepthEngine(0) as eng:
            eng.set_params(window_size=W, min_mapq=1, min_cov=4, step=10_000)
            eng.set_contigs(lengths)
            for t, r in reads.items():
                eng.push(t, r.pos, r.flag, r.mapq, r.cigar_off, r.cigar)
            eng.set_export(buf.data_ptr(), max_w, cap_b)
            eng.compute()
            torch.cuda.synchronize()
            h = buf.cpu().numpy()
            pb, nb = eng.device_runs()
            assert int(h[0]) == nb                               # the TRUE count, even past the capacity
            off = 0
            allb = []
            for t, l in enumerate(lengths):
                s, m = eng.windows(t)
                assert np.array_equal(h[1 + off:1 + off + nw[t]], s)
                assert np.array_equal(h[1 + max_w:1 + max_w + words_m].view(np.int32)[off:off + nw[t]], m)
                off += nw[t]
                d = po.perbase_c(reads.get(t, H.empty_reads()), 1, 0, l)
                r = H.oracle_runs(d, 4, 0, 10_000)
                allb.append(np.stack([r[:, 0], r[:, 2] | (t << 2)], 1))
            want = np.concatenate(allb).astype(np.int32)
            assert nb == len(want)
            k = min(nb, cap_b)
            got = h[1 + max_w + words_m:1 + max_w + words_m + k].view(np.int32).reshape(-1, 2)
            assert np.array_equal(got, want[:k])
            # fewer window slots than the job has windows: refused, not truncated
            eng.set_export(buf.data_ptr(), sum(nw) - 1, cap_b)
            with pytest.raises(GdError):
                eng.compute()
            eng.set_export(0, 0, 0)
            eng.compute()


def test_export_attached_before_the_first_compute_with_more_runs_than_the_initial_capacity():
    """gd_set_export before any compute, a job with more class boundaries than the boundary arrays hold at first
    (65 536): the first attempt's export must not read past the ordered array (it is clamped to its capacity);
    gd_compute grows the arrays, runs again, and the block is complete."""
    import torch
    from goleft_amd import shard
    from goleft_amd.engine import DepthEngine
    L = 400_000
    pos = np.arange(0, 300_000, 4, dtype=np.int32)                 # 75 000 one-base reads: 150 000 class changes
    n = len(pos)
    r = po.Reads(pos, np.zeros(n, np.uint16), np.full(n, 60, np.uint8), np.arange(n + 1, dtype=np.uint32),
                 np.full(n, (1 << 4) | 0, np.uint32))
    W = 1000
    nw = shard.n_windows(L, W)
    cap_b = 1 << 18
    words_m = (nw + 1) // 2
    dev = torch.device("cuda", 0)
    buf = torch.full((1 + nw + words_m + cap_b,), -1, dtype=torch.int64, device=dev)
    with DepthEngine(0) as eng:
        eng.set_params(window_size=W, min_mapq=1, min_cov=4)
        eng.set_contigs([L])
        eng.push(0, r.pos, r.flag, r.mapq, r.cigar_off, r.cigar)
        eng.set_export(buf.data_ptr(), nw, cap_b)
        eng.compute()
        assert eng.stats().reruns >= 1
        torch.cuda.synchronize()
        h = buf.cpu().numpy()
        d = po.perbase_c(r, 1, 0, L)
        want = H.oracle_runs(d, 4, 0, po.step_for(W))
        assert int(h[0]) == len(want) > 65536
        got = h[1 + nw + words_m:1 + nw + words_m + len(want)].view(np.int32).reshape(-1, 2)
        assert np.array_equal(got, np.stack([want[:, 0], want[:, 2]], 1))


def test_native_gather_of_the_export_block_world_of_one():
    """gd_comm_init / gd_gather_export: the exchange step inside the C ABI -- RCCL opened by
    the library, a grouped send / receive of the export block on the context's copy stream.  One GPU here: a world of
    one rank (the root sends to itself), which exercises the loading of RCCL, the communicator, the stream order
    between compute, gather and the next compute, two alternating buffers, and the state machine's refusals."""
    import torch
    from goleft_amd import shard, synth
    from goleft_amd.engine import DepthEngine, GdError, comm_unique_id
    L = 3_000_000
    r = po.Reads(*synth.short_reads_numpy(L, synth.n_reads_for(L, 12.0), 4))
    W = 1000
    dev = torch.device("cuda", 0)
    with DepthEngine(0) as eng:
        eng.set_params(window_size=W, min_mapq=1, min_cov=4)
        eng.set_contigs([L])
        eng.push(0, r.pos, r.flag, r.mapq, r.cigar_off, r.cigar)
        with pytest.raises(GdError) as ei:
            eng.gather_export(words=8)                         # no communicator yet
        assert ei.value.status == -4
        uid = comm_unique_id()
        assert len(uid) == 128
        eng.comm_init(0, 1, uid)
        with pytest.raises(GdError):
            eng.comm_init(0, 1, uid)                           # one communicator per context
        eng.compute()
        g = shard.RootGather([[0]], [L], W, 0, 1, dev, bounds_cap=0, native=True)
        g.reserve(eng.device_runs()[1])
        g.attach(eng)
        d = po.perbase_c(r, 1, 0, L)
        sums, mins = H.oracle_windows(d, W)
        for step in range(5):                                  # the buffers alternate; every step is a full compute + gather
            eng.compute()
            g.step_exported()
            res = g.result()
            assert not res["overflow"]
            parts = shard.unpack_gathered(res)
            assert np.array_equal(parts[0]["sums"].cpu().numpy(), sums), step
            assert np.array_equal(parts[0]["mins"].cpu().numpy(), mins), step
            runs = eng.callable_runs(0)
            assert np.array_equal(parts[0]["bounds"][:, 0].cpu().numpy(), runs[:, 0]), step
        # pipelined as bench.py does: finish(k - 1); flip(); launch(k); post()
        eng.compute_launch()
        for step in range(4):
            eng.compute_finish()
            g.flip()
            eng.compute_launch()
            g.post()
        eng.compute_finish()
        g.flip()
        g.post()
        g.drain()
        parts = shard.unpack_gathered(g.result())
        assert np.array_equal(parts[0]["sums"].cpu().numpy(), sums)
        # the same exchange with buffers from the ABI itself (what a cgo host has: no HIP binding of its own)
        import ctypes as C
        lib = eng._lib
        words = g.total
        ps, pr = C.c_void_p(), C.c_void_p()
        assert lib.gd_device_alloc(eng._ctx, 8 * words, C.byref(ps)) == 0 and lib.gd_device_alloc(eng._ctx, 8 * words, C.byref(pr)) == 0
        eng.set_export(ps.value, g.max_w, g.cap_b)
        eng.compute()
        eng.gather_export(recv_ptr=pr.value, words=words, root=0)
        got = np.empty(words, np.int64)
        assert lib.gd_device_read(eng._ctx, got.ctypes.data, pr, 8 * words) == 0
        assert got[0] == len(eng.callable_runs(0)) and np.array_equal(got[1:1 + len(sums)], sums)
        eng.set_export(0, 0, 0)
        assert lib.gd_device_free(eng._ctx, ps) == 0 and lib.gd_device_free(eng._ctx, pr) == 0
        with pytest.raises(GdError):
            eng.gather_export(words=10 ** 12)                  # more than the export block holds
        eng.comm_destroy()
        eng.comm_destroy()                                     # idempotent
