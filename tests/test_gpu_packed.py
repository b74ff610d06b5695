"""-m gpu: the packed-descriptor tile kernels (gd_tile_v8.hpp, GOLEFT_GD_KERNEL=v8) vs the CPU oracle and
vs the default v7 kernels on the same records -- every descriptor class (inlined single op, near complex, far
complex, not packable), both ways the descriptors get built (first gd_compute after gd_commit,
gd_adopt_device), bit exact."""
import ctypes
import os

import numpy as np
import pytest

from tests import helpers as H
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu

TILE = 1


def _engine(kernel="v8"):
    from goleft_amd.engine import DepthEngine
    old = os.environ.get("GOLEFT_GD_KERNEL")
    if kernel:
        os.environ["GOLEFT_GD_KERNEL"] = kernel      # read by gd_create
    try:
        e = DepthEngine(0)
    finally:
        if kernel:
            if old is None:
                del os.environ["GOLEFT_GD_KERNEL"]
            else:
                os.environ["GOLEFT_GD_KERNEL"] = old
    e.set_path(TILE)
    return e


def _results(e, contigs):
    out = []
    for tid in range(len(contigs)):
        s, m = e.windows(tid)
        out.append((e.perbase(tid), s, m, e.callable_runs(tid)))
    return out


def _run(e, contigs, reads, W=250, Q=1, mincov=4, maxmean=0, step=None):
    kw = dict(window_size=W, min_mapq=Q, min_cov=mincov, max_mean_depth=maxmean)
    if step:
        kw["step"] = step
    e.set_params(**kw)
    e.set_contigs([c[1] for c in contigs])
    for tid, r in reads.items():
        e.push(tid, r.pos, r.flag, r.mapq, r.cigar_off, r.cigar)
    e.compute()


def _check_oracle(e, contigs, reads, W, Q, mincov, maxmean, step=None):
    step = step or po.step_for(W)
    for tid, (_, clen) in enumerate(contigs):
        r = reads.get(tid, H.empty_reads())
        want = po.perbase_c(r, Q, 0, clen)
        got = e.perbase(tid)
        assert np.array_equal(got, want), "per-base tid %d first at %d" % (tid, int(np.nonzero(got != want)[0][0]))
        ws, wm = H.oracle_windows(want, W)
        gs, gm = e.windows(tid)
        assert np.array_equal(gs, ws) and np.array_equal(gm, wm)
        assert np.array_equal(e.callable_runs(tid), H.oracle_runs(want, mincov, maxmean, step))


def _mixed_reads(rng, length, n, p_complex, max_ops=8, big=False):
    """Short-read shaped stream: mostly one M op, a fraction with several ops; sprinkled with the
    descriptor corner cases (no CIGAR, zero-length op, single = / X / S / D op, a single M longer
    than the 12-bit inline field, more than 32 ops)."""
    pos = np.sort(rng.integers(0, max(1, length), size=n)).astype(np.int32)
    cig, off = [], [0]
    for i in range(n):
        u = rng.random()
        if u < p_complex:
            k = int(rng.integers(2, max_ops + 1))
            if big and rng.random() < 0.05:
                k = int(rng.integers(33, 80))
            ops = rng.choice(9, size=k, p=[0.5, 0.12, 0.12, 0.04, 0.1, 0.02, 0.02, 0.04, 0.04])
            lens = rng.integers(0, 90, size=k)
        elif u < p_complex + 0.03:
            kind = int(rng.integers(0, 7))
            if kind == 0:
                ops, lens = np.zeros(0, int), np.zeros(0, int)                 # no CIGAR
            elif kind == 1:
                ops, lens = np.array([0]), np.array([0])                        # 0M
            elif kind == 2:
                ops, lens = np.array([7]), np.array([int(rng.integers(1, 200))])   # =
            elif kind == 3:
                ops, lens = np.array([8]), np.array([int(rng.integers(1, 200))])   # X
            elif kind == 4:
                ops, lens = np.array([4]), np.array([int(rng.integers(1, 200))])   # S only
            elif kind == 5:
                ops, lens = np.array([2]), np.array([int(rng.integers(1, 200))])   # D only
            else:
                ops, lens = np.array([0]), np.array([int(rng.integers(4094, 9000))])   # around the inline limit
        else:
            ops, lens = np.array([0]), np.array([int(rng.integers(30, 152))])
        cig.extend(((np.asarray(lens, np.uint32) << 4) | np.asarray(ops, np.uint32)).tolist())
        off.append(len(cig))
    flag = rng.choice([0, 16, 99, 147, 0x400, 0x100, 0x200, 0x4, 0x800, 0x410, 0x1, 0xa3],
                      size=n).astype(np.uint16)
    mapq = rng.choice([0, 1, 5, 60, 255], size=n).astype(np.uint8)
    return po.Reads(pos, flag, mapq, np.asarray(off, np.uint32), np.asarray(cig, np.uint32))


@pytest.mark.parametrize("seed,p_complex,big", [(0, 0.05, False), (1, 0.3, False), (2, 0.95, False),
                                                (3, 0.6, True), (4, 0.0, False)])
def test_packed_descriptors_against_oracle_and_v7(seed, p_complex, big):
    rng = np.random.default_rng(100 + seed)
    lens = [70001, 4096, 1, 23456, 8193]
    contigs = [("p%d" % i, l) for i, l in enumerate(lens)]
    reads = {}
    for tid, l in enumerate(lens):
        if tid == 2:
            continue
        reads[tid] = _mixed_reads(rng, l, int(rng.integers(50, 9000)), p_complex, big=big)
    W = int(rng.choice([1, 13, 100, 250, 1000]))
    Q = int(rng.choice([0, 1, 20]))
    mincov, maxmean = int(rng.integers(1, 6)), int(rng.choice([0, 25]))
    e8, e7 = _engine("v8"), _engine("v7")
    try:
        for e in (e8, e7):
            _run(e, contigs, reads, W, Q, mincov, maxmean)
        _check_oracle(e8, contigs, reads, W, Q, mincov, maxmean)
        for a, b in zip(_results(e8, contigs), _results(e7, contigs)):
            for x, y in zip(a, b):
                assert np.array_equal(x, y)
        s8, s7 = e8.stats(), e7.stats()
        assert (s8.max_span_seen, s8.lookback, s8.n_runs, s8.path) == (s7.max_span_seen, s7.lookback, s7.n_runs, s7.path)
        # a second compute with another filter reuses the descriptors
        e8.set_params(window_size=W, min_mapq=60, min_cov=mincov, max_mean_depth=maxmean)
        e8.compute()
        _check_oracle(e8, contigs, reads, W, 60, mincov, maxmean)
    finally:
        e8.close()
        e7.close()


def test_dense_complex_units_and_deep_tiles():
    """Every read complex and thousands of them on one tile: deltas overflow into `far`
    descriptors, batches overflow the per-wave queue, several record batches per tile."""
    rng = np.random.default_rng(7)
    L = 20000
    n = 30000
    pos = np.sort(rng.integers(3000, 5000, size=n)).astype(np.int32)
    k = rng.integers(2, 6, size=n)
    off = np.zeros(n + 1, np.uint32)
    off[1:] = np.cumsum(k)
    m = int(off[-1])
    ops = rng.choice([0, 1, 2, 4], size=m, p=[0.6, 0.15, 0.15, 0.1])
    lens = rng.integers(1, 60, size=m)
    r = po.Reads(pos, np.zeros(n, np.uint16), np.full(n, 60, np.uint8), off,
                 ((lens.astype(np.uint32) << 4) | ops.astype(np.uint32)).astype(np.uint32))
    contigs = [("d", L)]
    e = _engine()
    try:
        _run(e, contigs, {0: r}, 100, 1, 4, 0)
        _check_oracle(e, contigs, {0: r}, 100, 1, 4, 0)
    finally:
        e.close()


def test_unpackable_flags_take_the_previous_kernel():
    """A FLAG bit above 0xfff does not fit the descriptor: the contig is not packed, results stay exact
    (flag_mask 0x1704 filters on the high bit)."""
    rng = np.random.default_rng(9)
    L = 50000
    r = _mixed_reads(rng, L, 4000, 0.1)
    r.flag[::7] |= 0x1000
    contigs = [("u", L)]
    e = _engine()
    try:
        e.set_params(window_size=250, min_mapq=1, min_cov=4, flag_mask=0x1704)
        e.set_contigs([L])
        e.push(0, r.pos, r.flag, r.mapq, r.cigar_off, r.cigar)
        e.compute()
        want = po.perbase_c(r, 1, 0, L, flag_mask=0x1704)
        assert not np.array_equal(want, po.perbase_c(r, 1, 0, L))      # the high bit does filter reads
        assert np.array_equal(e.perbase(0), want)
    finally:
        e.close()


class _DevArray:
    """numpy array copied to HBM with the HIP runtime; quacks like the tensors adopt_device takes."""
    _hip = None

    def __init__(self, a):
        if _DevArray._hip is None:
            hip = ctypes.CDLL("libamdhip64.so")
            hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
            hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
            hip.hipFree.argtypes = [ctypes.c_void_p]
            _DevArray._hip = hip
        a = np.ascontiguousarray(a)
        self.shape, self._size, self.is_cuda = a.shape, a.itemsize, True
        self._p = ctypes.c_void_p()
        assert self._hip.hipMalloc(ctypes.byref(self._p), max(a.nbytes, 1)) == 0
        if a.nbytes:
            assert self._hip.hipMemcpy(self._p, a.ctypes.data, a.nbytes, 1) == 0     # host -> device

    def is_contiguous(self):
        return True

    def element_size(self):
        return self._size

    def data_ptr(self):
        return self._p.value

    def free(self):
        self._hip.hipFree(self._p)


def test_adopted_device_records_are_packed_on_arrival():
    rng = np.random.default_rng(11)
    L = 300000
    r = _mixed_reads(rng, L, 60000, 0.08)
    contigs = [("a", L)]
    e = _engine()
    dev = [_DevArray(x) for x in (r.pos, r.flag, r.mapq, r.cigar_off, r.cigar)]
    try:
        from goleft_amd import engine as E
        e.set_params(window_size=1000, min_mapq=1, min_cov=4)
        e.set_contigs([L])
        e.set_profiling(True)
        e.adopt_device(0, *dev)
        assert e.kernel_ms(E.K_PACK) > 0.0           # built by gd_adopt_device ...
        packed_ms = e.kernel_ms(E.K_PACK)
        e.compute()
        assert e.kernel_ms(E.K_PACK) == packed_ms    # ... not by gd_compute
        _check_oracle(e, contigs, {0: r}, 1000, 1, 4, 0)
        # sums-only output on the same descriptors
        e.set_outputs(perbase=False, sums_only=True)
        e.compute()
        want = po.perbase_c(r, 1, 0, L)
        assert np.array_equal(e.window_sums(0), H.oracle_windows(want, 1000)[0])
    finally:
        e.close()
        for d in dev:
            d.free()
