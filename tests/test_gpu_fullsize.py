"""-m gpu: BASELINE.json config 2 at its FULL size (synthetic 30x chr20, 63 025 520 bp,
12.6 M reads) and the long-read path on a full chr20 (20x ONT-like, ~10^8 CIGAR ops), on the
device paths that serve them.  Checked two ways: directly against the C oracle (which still
finishes a chromosome in about a second), and through size-independent properties that need
no oracle at all -- conservation of counted bases, window sums / minima recomputed from the
per-base vector, class runs tiling the contig with breaks exactly at class changes and at
multiples of the step."""
import os

import numpy as np
import pytest

from goleft_amd import synth
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu

L = synth.CHR20_LEN
W, Q, MINCOV = 1000, 1, 4


def counted_bases(r, q, length, flag_mask=0x704):
    """sum over kept reads of the M/=/X bases that fall inside [0, length): what the per-base
    vector must add up to (`samtools depth` counts nothing else, depth/depth.go:45)."""
    n = r.n
    keep = ((r.flag & flag_mask) == 0) & (r.mapq >= q)
    op = (r.cigar & 0xF).astype(np.int64)
    ln = (r.cigar >> 4).astype(np.int64)
    consumes = np.isin(op, (0, 2, 3, 7, 8))
    counted = np.isin(op, (0, 7, 8))
    read_of = np.repeat(np.arange(n), np.diff(r.cigar_off.astype(np.int64)))
    cons = np.where(consumes, ln, 0)
    before = np.cumsum(cons) - cons                           # consumed before this op, global
    first = r.cigar_off[:-1].astype(np.int64)
    base = np.where(np.diff(r.cigar_off.astype(np.int64)) > 0, before[np.minimum(first, len(before) - 1)], 0)
    start = r.pos.astype(np.int64)[read_of] + before - base[read_of]
    end = np.minimum(start + ln, length)
    return int(np.where(counted & keep[read_of], np.maximum(end - np.maximum(start, 0), 0), 0).sum())


def check_properties(eng, r, q, perbase):
    assert perbase.shape == (L,) and perbase.min() >= 0
    assert int(perbase.sum(dtype=np.int64)) == counted_bases(r, q, L)
    sums, mins = eng.windows(0)
    edges = np.arange(0, L, W)
    assert np.array_equal(sums, np.add.reduceat(perbase.astype(np.int64), edges))
    assert np.array_equal(mins, np.minimum.reduceat(perbase, edges))
    runs = eng.callable_runs(0)
    step = po.step_for(W)
    assert runs[0, 0] == 0 and runs[-1, 1] == L and np.array_equal(runs[1:, 0], runs[:-1, 1])
    cls = np.where(perbase == 0, 0, np.where(perbase < MINCOV, 1, 2)).astype(np.int8)
    brk = np.flatnonzero((cls[1:] != cls[:-1]) | (np.arange(1, L) % step == 0)) + 1
    assert np.array_equal(runs[1:, 0], brk)                   # breaks exactly there, nowhere else
    assert np.array_equal(runs[:, 2], cls[runs[:, 0]])


def test_config2_chr20_full_size():
    from goleft_amd.engine import DepthEngine, PATH_TILE
    n = synth.n_reads_for(L)
    assert n == 12605104
    r = po.Reads(*synth.short_reads_numpy(L, n, 20))
    with DepthEngine(0) as eng:
        eng.set_params(window_size=W, min_mapq=Q, min_cov=MINCOV)
        eng.set_path(PATH_TILE)
        eng.set_contigs([L])
        eng.push(0, r.pos, r.flag, r.mapq, r.cigar_off, r.cigar)
        eng.compute()
        got = eng.perbase(0)
        check_properties(eng, r, Q, got)
    assert np.array_equal(got, po.perbase_c(r, Q, 0, L, diff=True))


def test_ont_chr20_full_size_chunk_path():
    from goleft_amd.engine import DepthEngine, PATH_CHUNK
    n = synth.n_ont_reads_for(L, 20.0)
    r = po.Reads(*synth.ont_reads_numpy(L, n, 20))
    assert r.cigar.shape[0] > 50_000_000
    with DepthEngine(0) as eng:
        eng.set_params(window_size=W, min_mapq=Q, min_cov=MINCOV)
        eng.set_path(PATH_CHUNK)
        eng.set_contigs([L])
        eng.push(0, r.pos, r.flag, r.mapq, r.cigar_off, r.cigar)
        eng.compute()
        got = eng.perbase(0)
        check_properties(eng, r, Q, got)
    assert np.array_equal(got, po.perbase_c(r, Q, 0, L, diff=True))


# ---------------------------------------------------------------------------------------------------------
# BASELINE.json's FULL sizes (config 3: 30x WGS, 3.1 Gb; config 5: 20x ONT-like WGS), checked through
# size-independent properties computed ON THE DEVICE with torch from the record streams and the engine's own
# per-base vector (the bit-for-bit comparison with the oracle, every contig, all host cores, follows below):
#   * conservation: every contig's per-base vector adds up to the M/=/X bases of its kept reads that fall inside
#     the contig (computed from the records alone, independent of any engine structure);
#   * window sums / minima equal reductions of the per-base vector;
#   * class runs tile every contig, with breaks exactly at class changes and at multiples of the step.
# ---------------------------------------------------------------------------------------------------------
def _counted_bases_torch(torch, pos, flag, mapq, off, cig, q, length, flag_mask=0x704):
    keep = ((flag.to(torch.int32) & flag_mask) == 0) & (mapq.to(torch.int32) >= q)
    op = (cig & 0xF).to(torch.int64)
    ln = ((cig.to(torch.int64) & 0xFFFFFFFF) >> 4)
    consumes = (op == 0) | (op == 2) | (op == 3) | (op == 7) | (op == 8)
    counted = (op == 0) | (op == 7) | (op == 8)
    o = off.to(torch.int64) & 0xFFFFFFFF
    nops = o[1:] - o[:-1]
    read_of = torch.repeat_interleave(torch.arange(nops.numel(), device=pos.device), nops)
    cons = torch.where(consumes, ln, torch.zeros_like(ln))
    before = torch.cumsum(cons, 0) - cons
    first = torch.clamp(o[:-1], max=max(0, cig.numel() - 1))
    base = torch.where(nops > 0, before[first], torch.zeros_like(first))
    start = pos.to(torch.int64)[read_of] + before - base[read_of]
    end = torch.clamp(start + ln, max=length)
    c = torch.where(counted & keep[read_of], torch.clamp(end - torch.clamp(start, min=0), min=0), torch.zeros_like(ln))
    return int(c.sum().item())


def _check_genome_properties(torch, eng, dev, lengths, streams, step):
    from goleft_amd import shard
    ps, pm, nwt = eng.device_windows()
    sums_all = shard.device_view(ps, nwt, torch.int64, dev)
    mins_all = shard.device_view(pm, nwt, torch.int32, dev)
    pb, nb = eng.device_runs()
    bounds = shard.device_view(pb, 2 * nb, torch.int32, dev).view(-1, 2)
    ctg_of = bounds[:, 1] >> 2
    for t, Lt in enumerate(lengths):
        p, n = eng.device_perbase(t)
        d = shard.device_view(p, n, torch.int32, dev)
        assert n == Lt and int(d.min().item()) >= 0
        total = int(d.sum(dtype=torch.int64).item())
        assert total == _counted_bases_torch(torch, *streams[t], Q, Lt), t          # conservation
        o, nw = eng.window_offset(t)
        pad = (-Lt) % W
        dd = torch.cat([d.to(torch.int64), torch.zeros(pad, dtype=torch.int64, device=dev)]).view(-1, W)
        assert torch.equal(sums_all[o:o + nw], dd.sum(1)), t
        dm = torch.cat([d, torch.full((pad,), 2 ** 31 - 1, dtype=torch.int32, device=dev)]).view(-1, W)
        assert torch.equal(mins_all[o:o + nw], dm.min(1).values), t
        del dd, dm
        cls = torch.where(d == 0, 0, torch.where(d < MINCOV, 1, 2)).to(torch.int8)
        ar = torch.arange(1, Lt, device=dev)
        brk = torch.nonzero((cls[1:] != cls[:-1]) | (ar % step == 0)).flatten() + 1
        b = bounds[ctg_of == t]
        assert int(b[0, 0].item()) == 0                                              # the first run starts the contig
        assert torch.equal(b[1:, 0].to(torch.int64), brk), t                         # breaks exactly there, nowhere else
        assert torch.equal((b[:, 1] & 3).to(torch.int8), cls[b[:, 0].to(torch.int64)]), t
        del cls, ar, brk


def test_config3_wgs_full_size_properties():
    """30x WGS, hg19 contig lengths, 619 M reads: the configuration bench.py's headline is quoted on."""
    import torch
    from goleft_amd.engine import DepthEngine
    dev = torch.device("cuda", 0)
    lengths = list(synth.HG19_LENGTHS)
    assert sum(lengths) == 3_095_677_412
    with DepthEngine(0) as eng:
        eng.set_params(window_size=W, min_mapq=Q, min_cov=MINCOV)
        eng.set_contigs(lengths)
        streams = []
        for t, Lt in enumerate(lengths):
            s = synth.short_reads_torch(Lt, synth.n_reads_for(Lt), t + 1, dev)
            streams.append(s)
            eng.adopt_device(t, *s)
        eng.compute()
        st = eng.stats()
        assert st.n_reads == 619_135_482 and st.path == 1 and st.reruns == 0
        _check_genome_properties(torch, eng, dev, lengths, streams, po.step_for(W))


def test_config5_ont_wgs_full_size_properties():
    """20x ONT-like WGS (4.8e9 CIGAR ops): the long-read path at BASELINE.json's config-5 size."""
    import torch
    from goleft_amd.engine import DepthEngine
    dev = torch.device("cuda", 0)
    lengths = list(synth.HG19_LENGTHS)
    with DepthEngine(0) as eng:
        eng.set_params(window_size=W, min_mapq=Q, min_cov=MINCOV)
        eng.set_contigs(lengths)
        streams = []
        for t, Lt in enumerate(lengths):
            s = synth.ont_reads_torch(Lt, synth.n_ont_reads_for(Lt, 20.0), t + 1, dev)
            streams.append(s)
            eng.adopt_device(t, *s)
        eng.compute()
        assert eng.stats().path == 3                                                 # GD_PATH_CHUNK chosen by AUTO
        _check_genome_properties(torch, eng, dev, lengths, streams, po.step_for(W))


def _bit_compare_genome(eng, lengths, streams, Wd, q, mincov, maxmean=0):
    """EVERY contig of a resident genome against the C oracle, bit for bit: the per-base vector (compared tile by
    tile as the oracle produces it, oracle/pyoracle.py::tiled_contig_check, all host cores), and window sums /
    minima / class runs derived from the ORACLE's vector -- nothing here reads the engine's own per-base output
    to judge its other outputs."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    cores = os.cpu_count() or 8
    outer = 4                                                  # contigs in flight (their host-side numpy passes overlap)
    step = po.step_for(Wd)

    def one(t):
        a = [x.cpu().numpy() for x in streams[t]]
        r = po.Reads(a[0], a[1].view(np.uint16), a[2], a[3].view(np.uint32), a[4].view(np.uint32))
        got = eng.perbase(t)
        res = po.tiled_contig_check(r, q, lengths[t], Wd, mincov, maxmean, step, max(2, cores // outer), got=got)
        assert res["equal"], "contig %d: per-base depth differs from the oracle, first at %d" % (t, res["first_diff"])
        sums, mins = eng.windows(t)
        assert np.array_equal(sums, res["sums"]), "contig %d: window sums" % t
        assert np.array_equal(mins, res["mins"]), "contig %d: window minima" % t
        runs = eng.callable_runs(t)
        assert np.array_equal(runs[:, 0], res["run_starts"]), "contig %d: run starts" % t
        assert np.array_equal(runs[:, 2], res["run_cls"]), "contig %d: run classes" % t
        assert runs[-1, 1] == lengths[t] and np.array_equal(runs[1:, 0], runs[:-1, 1])
        return lengths[t]

    with ThreadPoolExecutor(max_workers=outer) as ex:
        assert sum(ex.map(one, range(len(lengths)))) == sum(lengths)


def test_config3_wgs_full_size_bit_exact():
    """BASELINE.json config 3 at its full size, every contig compared with the C oracle bit for bit (30x WGS, hg19
    lengths, 619 M reads): what bench.py's headline is quoted on, through the kernel the default configuration
    runs: the straight-line tile kernel on the records as they arrived."""
    import torch
    from goleft_amd import engine as E
    dev = torch.device("cuda", 0)
    lengths = list(synth.HG19_LENGTHS)
    with E.DepthEngine(0) as eng:
        eng.set_params(window_size=W, min_mapq=Q, min_cov=MINCOV)
        eng.set_contigs(lengths)
        streams = []
        for t, Lt in enumerate(lengths):
            s = synth.short_reads_torch(Lt, synth.n_reads_for(Lt), t + 1, dev)
            streams.append(s)
            eng.adopt_device(t, *s)
        eng.compute()
        st = eng.stats()
        assert st.n_reads == 619_135_482 and st.tile_kernel == E.TK_FAST_RAW
        assert st.n_slow_tiles < 100 and st.reruns == 0 and st.lookback == 192   # (of 755 785; the look-back was measured as the records arrived)
        _bit_compare_genome(eng, lengths, streams, W, Q, MINCOV)


def test_config5_ont_wgs_full_size_bit_exact():
    """BASELINE.json config 5 at its full size (20x ONT-like WGS, 4.8e9 CIGAR ops), every contig compared with the
    C oracle bit for bit."""
    import torch
    from goleft_amd.engine import DepthEngine
    dev = torch.device("cuda", 0)
    lengths = list(synth.HG19_LENGTHS)
    with DepthEngine(0) as eng:
        eng.set_params(window_size=W, min_mapq=Q, min_cov=MINCOV)
        eng.set_contigs(lengths)
        streams = []
        for t, Lt in enumerate(lengths):
            s = synth.ont_reads_torch(Lt, synth.n_ont_reads_for(Lt, 20.0), t + 1, dev)
            streams.append(s)
            eng.adopt_device(t, *s)
        eng.compute()
        assert eng.stats().path == 3
        _bit_compare_genome(eng, lengths, streams, W, Q, MINCOV)


def test_config4_cohort_full_size_against_the_oracle():
    """200 samples x chr1 (BASELINE.json config 4), window sums only (what the depthwed matrix is made of), through
    the streaming sums kernel.  Every sample's window sums add up to the M/=/X bases of its kept reads inside the contig
    (from the records alone); and for FIVE samples drawn at random the window sums are compared, bit for bit, with the
    C oracle's per-base depth of the same records (oracle/depth_oracle.c in 10 Mb tiles, reduced per window), and their
    columns of the sites x samples matrix (gd_depthwed_device at -s 1000) with the restatement of the reference's text
    chain -- "%.4g" of sum / length per row (depth/depth.go:301), int(0.5 + parse) summed over a group
    (depthwed/depthwed.go:93-157) -- computed FROM THE ORACLE'S sums: nothing of the engine judges the engine."""
    import torch
    from goleft_amd.engine import DepthEngine
    dev = torch.device("cuda", 0)
    L1, S, Wc, size = synth.HG19_LENGTHS[0], 200, 250, 1000
    rng = np.random.default_rng(404)
    picked = sorted(int(x) for x in rng.choice(S, size=5, replace=False))
    with DepthEngine(0) as eng:
        eng.set_params(window_size=Wc, min_mapq=Q, min_cov=MINCOV)
        eng.set_outputs(sums_only=True)
        eng.set_contigs([L1] * S)
        want = []
        keep = {}
        for s in range(S):
            st = synth.short_reads_torch(L1, synth.n_reads_for(L1), 1000 + s, dev)
            eng.adopt_device(s, *st)
            want.append(_counted_bases_torch(torch, *st, Q, L1))
            keep[s] = st                                       # adopted, not copied: the engine reads these tensors
        eng.compute()
        assert eng.stats().tile_kernel == 8                    # GD_TK_SUMS_STREAM_RAW: the streaming kernel on the records as they arrived
        from goleft_amd import shard
        ps, _, nwt = eng.device_windows()
        sums_all = shard.device_view(ps, nwt, torch.int64, dev)
        nw = (L1 + Wc - 1) // Wc
        got = sums_all.view(S, nw).sum(1).tolist()
        assert got == want
        mine = {s: sums_all.view(S, nw)[s].cpu().numpy() for s in picked}
        tids = np.asarray(picked, np.int32).reshape(-1, 1)
        ptr, rows = eng.depthwed_device(tids, size)
        matrix = shard.device_view(ptr, rows * len(picked), torch.int64, dev).view(rows, len(picked)).cpu().numpy()
        cores = os.cpu_count() or 8
        for j, s in enumerate(picked):
            a = [x.cpu().numpy() for x in keep[s]]
            rd = po.Reads(a[0], a[1].view(np.uint16), a[2], a[3].view(np.uint32), a[4].view(np.uint32))
            res = po.tiled_contig_check(rd, Q, L1, Wc, MINCOV, 0, po.step_for(Wc), cores)     # the oracle's vector, tile by tile
            assert np.array_equal(mine[s], res["sums"]), "sample %d: window sums differ from the oracle" % s
            cells, st_, en_ = po.depthwed_cells_contig(res["sums"], L1, Wc, size)
            assert len(cells) == rows and st_[0] == 0 and en_[-1] == L1
            assert np.array_equal(matrix[:, j], cells), "sample %d: matrix column differs" % s
            del a, rd, res


@pytest.mark.parametrize("path,mode", [(1, "full"), (1, "windows"), (1, "sums"), (3, "full"), (3, "windows"),
                                       (2, "full"), (0, "full")])
def test_maximum_contig_length(path, mode):
    """GD_MAX_CONTIG_LENGTH (0x7fff0000 positions: 524 272 tiles, an 8.6 GB per-base vector): records at the
    start, across the tile boundary in the middle, and at the very end -- reads that hang over the contig end,
    a deletion and a skip that run past it, a read at the last position -- on every device algorithm and
    output mode.  Per-base values and window reductions of the three busy stretches against the oracle,
    everything else zero / NO_COVERAGE, and one position past the limit is GD_E_RANGE."""
    from goleft_amd.engine import DepthEngine, GdError
    from tests import helpers as H
    LMAX = 0x7fff0000
    rng = np.random.default_rng(8)
    spots = [0, LMAX // 2 - 20000, LMAX - 70000]
    parts = []
    for s0 in spots:
        r = H.random_reads(rng, 60000, 3000, max_len=400, long_reads=(path != 1))
        parts.append((r.pos.astype(np.int64) + s0, r))
    pos = np.concatenate([p for p, _ in parts] + [[LMAX - 5000, LMAX - 300, LMAX - 1, LMAX - 1]])
    tail_cig = np.array([(3000 << 4) | 0, (900 << 4) | 2, (2000 << 4) | 0,          # M D M over the end
                         (100 << 4) | 0, (50000 << 4) | 3, (10 << 4) | 0,           # M N(past the end) M
                         (1 << 4) | 0, (77 << 4) | 0], np.uint32)
    tail_off = np.array([3, 6, 7, 8])
    flag = np.concatenate([r.flag for _, r in parts] + [np.zeros(4, np.uint16)])
    mapq = np.concatenate([r.mapq for _, r in parts] + [np.full(4, 60, np.uint8)])
    offs, cigs, base = [np.zeros(1, np.int64)], [], 0
    for _, r in parts:
        offs.append(r.cigar_off[1:].astype(np.int64) + base)
        cigs.append(r.cigar)
        base += int(r.cigar_off[-1])
    offs.append(tail_off + base)
    cigs.append(tail_cig)
    order = np.argsort(pos, kind="stable")                    # (the tail records already sort last)
    assert np.array_equal(order, np.arange(len(pos)))
    r = po.Reads(pos.astype(np.int32), flag.astype(np.uint16), mapq.astype(np.uint8),
                 np.concatenate(offs).astype(np.uint32), np.concatenate(cigs).astype(np.uint32))
    Wm, step = 1000, po.step_for(1000)
    with DepthEngine(0) as eng:
        with pytest.raises(GdError) as ei:
            eng.set_contigs([LMAX + 1])
        assert ei.value.status == -5
        eng.set_path(path)
        eng.set_outputs(perbase=(mode == "full"), sums_only=(mode == "sums"))
        eng.set_params(window_size=Wm, min_mapq=Q, min_cov=MINCOV)
        eng.set_contigs([LMAX])
        eng.push(0, r.pos, r.flag, r.mapq, r.cigar_off, r.cigar)
        eng.compute()
        sums = eng.window_sums(0)
        assert len(sums) == (LMAX + Wm - 1) // Wm
        busy = np.zeros(len(sums), bool)
        total = 0
        for s0 in spots:
            a, b = (s0 // Wm) * Wm, min(LMAX, ((s0 + 200000) // Wm + 1) * Wm)
            want = po.perbase_c(r, Q, a, b)
            total += int(want.sum(dtype=np.int64))
            ws, wm = H.oracle_windows(want, Wm, a)
            k0, k1 = a // Wm, a // Wm + len(ws)
            busy[k0:k1] = True
            assert np.array_equal(sums[k0:k1], ws), (s0, "sums")
            if mode == "sums":
                continue
            assert np.array_equal(eng.windows(0)[1][k0:k1], wm), (s0, "mins")
            if mode == "full":
                assert np.array_equal(eng.perbase(0, a, b), want), (s0, "perbase")
                assert np.array_equal(eng.region_callable(0, a, b), H.oracle_runs(want, MINCOV, 0, 1 << 62, a))
        assert not sums[~busy].any() and int(sums.sum()) == total == counted_bases(r, Q, LMAX)
        if mode != "sums":
            runs = eng.callable_runs(0)
            assert runs[0, 0] == 0 and runs[-1, 1] == LMAX and np.array_equal(runs[1:, 0], runs[:-1, 1])
            assert (np.diff(runs[:, 0] // step) <= 1).all()                   # no run crosses a step boundary
            far = (runs[:, 1] <= spots[1] - 40000) & (runs[:, 0] >= 250000)   # between the first two stretches
            assert far.any() and (runs[far, 2] == 0).all()
