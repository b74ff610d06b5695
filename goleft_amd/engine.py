"""DepthEngine -- thin Python handle on the C ABI (include/goleft_depth.h).

Host-side convenience used by the `depth` front end (depth.py), the tests and
bench.py.  All arithmetic happens in the HIP library; numpy is only the
container for inputs and results."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import _lib
from ._lib import GdBatch, GdParams, GdRun, GdStats

CLASS_NAMES = ("NO_COVERAGE", "LOW_COVERAGE", "CALLABLE", "EXCESSIVE_COVERAGE")
K_PREP, K_TILE, K_RUNS, K_EXPAND, K_SCAN, K_CKPT, K_SEQSTATS, K_MDFLAGS, K_INFLATE, K_NORM = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9
# gd_set_option keys (include/goleft_depth.h)
OPT_NT_STORES, OPT_FAST_KERNEL, OPT_COPY_THREADS, OPT_PUSH_THREADS, OPT_H2D_KERNEL, OPT_PUSH_CHUNK, OPT_BAM_REFS = 3, 5, 6, 7, 8, 9, 10
OPT_INGEST_INDEX = 14
OPT_COMMIT_CHECK = 23
PATH_AUTO, PATH_TILE, PATH_SCATTER, PATH_CHUNK = 0, 1, 2, 3
# gd_stats.tile_kernel (include/goleft_depth.h GD_TK_*)
TK_NONE, TK_GENERIC, TK_FAST, TK_FAST_RAW, TK_LONG, TK_SCATTER, TK_SUMS_STREAM, TK_TILE_SUMS, TK_SUMS_STREAM_RAW = range(9)
TK_NAMES = ("none", "gd_tile_kernel", "gd_tile_fast_kernel", "gd_tile_fast_kernel<raw>", "gd_ltile2_kernel",
            "gd_expand_scatter_kernel+gd_scan_kernel", "gd_sums_stream_kernel", "gd_tile_sums_kernel", "gd_sums_stream_kernel<raw>")


class GdError(RuntimeError):
    def __init__(self, status: int, msg: str):
        super().__init__("goleft_depth: %s (status %d)" % (msg, status))
        self.status = status


class DepthEngine:
    """One context == one HIP device."""

    def __init__(self, device: int = 0, stream: Optional[int] = None):
        self._lib = _lib.load()
        self._ctx = C.c_void_p()
        rc = self._lib.gd_create(device, C.byref(self._ctx))
        if rc != 0:
            raise GdError(rc, self._lib.gd_strerror(rc).decode())
        self.device = device
        self.contig_lengths: list[int] = []
        self._keep = []   # device tensors adopted by the engine must stay alive
        if stream is not None:
            self._chk(self._lib.gd_set_stream(self._ctx, C.c_void_p(stream)))

    # -- plumbing ---------------------------------------------------------
    def _chk(self, rc: int):
        if rc != 0:
            detail = self._lib.gd_last_error(self._ctx).decode()
            raise GdError(rc, "%s: %s" % (self._lib.gd_strerror(rc).decode(), detail))

    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx:
            self._lib.gd_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- configuration ----------------------------------------------------
    def set_params(self, window_size=250, min_mapq=1, min_cov=4, max_mean_depth=0,
                   flag_mask=0x704, max_span_hint=0, step=0):
        p = GdParams(window_size, min_mapq, min_cov, max_mean_depth, flag_mask, max_span_hint, step)
        self._chk(self._lib.gd_set_params(self._ctx, C.byref(p)))
        self.params = p

    def set_path(self, path: int):
        """PATH_AUTO / PATH_TILE / PATH_SCATTER / PATH_CHUNK (include/goleft_depth.h GD_PATH_*)."""
        self._chk(self._lib.gd_set_path(self._ctx, int(path)))

    def set_outputs(self, perbase: bool = True, sums_only: bool = False):
        """perbase=False: windows + class runs only (no 4 B/base vector in HBM);
        sums_only=True: window sums and nothing else (GD_OUT_SUMS_ONLY)."""
        self._chk(self._lib.gd_set_outputs(self._ctx, 2 if sums_only else (1 if perbase else 0)))

    def set_contigs(self, lengths: Sequence[int]):
        a = np.asarray(lengths, dtype=np.int64)
        self._chk(self._lib.gd_set_contigs(self._ctx, len(a), a.ctypes.data))
        self.contig_lengths = [int(x) for x in a]
        self._keep = []

    def select_contigs(self, tids: Sequence[int]):
        a = np.asarray(tids, dtype=np.int32)
        self._chk(self._lib.gd_select_contigs(self._ctx, len(a), a.ctypes.data if len(a) else None))

    # -- ingest -----------------------------------------------------------
    def push(self, tid: int, pos, flag, mapq, cigar_off, cigar):
        pos = np.ascontiguousarray(pos, np.int32)
        flag = np.ascontiguousarray(flag, np.uint16)
        mapq = np.ascontiguousarray(mapq, np.uint8)
        cigar_off = np.ascontiguousarray(cigar_off, np.uint32)
        cigar = np.ascontiguousarray(cigar, np.uint32)
        n = pos.shape[0]
        assert cigar_off.shape[0] == n + 1
        self._chk(self._lib.gd_push(self._ctx, tid, pos.ctypes.data, flag.ctypes.data,
                                    mapq.ctypes.data, cigar_off.ctypes.data,
                                    cigar.ctypes.data if cigar.shape[0] else None,
                                    n, cigar.shape[0]))

    def acquire(self, reads_cap: int, ops_cap: int) -> GdBatch:
        b = GdBatch()
        self._chk(self._lib.gd_acquire(self._ctx, reads_cap, ops_cap, C.byref(b)))
        return b

    def commit(self, b: GdBatch, tid: int, n_reads: int, n_ops: int):
        self._chk(self._lib.gd_commit(self._ctx, C.byref(b), tid, n_reads, n_ops))

    def adopt_device(self, tid: int, pos, flag, mapq, cigar_off, cigar):
        """Zero-copy: torch device tensors (int32/int16/uint8/int32/int32 storage)."""
        n = int(pos.shape[0])
        m = int(cigar.shape[0])
        assert int(cigar_off.shape[0]) == n + 1
        for t, sz in ((pos, 4), (flag, 2), (mapq, 1), (cigar_off, 4), (cigar, 4)):
            assert t.is_cuda and t.is_contiguous() and t.element_size() == sz
        b = GdBatch(pos.data_ptr(), flag.data_ptr(), mapq.data_ptr(), cigar_off.data_ptr(),
                    cigar.data_ptr(), n, m, -1)
        self._chk(self._lib.gd_adopt_device(self._ctx, tid, C.byref(b), n, m))
        self._keep.append((pos, flag, mapq, cigar_off, cigar))

    def reset(self):
        self._chk(self._lib.gd_reset(self._ctx))
        self._keep = []

    # -- compute ----------------------------------------------------------
    def compute(self):
        self._chk(self._lib.gd_compute(self._ctx))

    def drop_derived(self):
        """gd_drop_derived: back to the state right after the records arrived."""
        self._chk(self._lib.gd_drop_derived(self._ctx))

    def rebuild_derived(self):
        """gd_rebuild_derived: every derived structure the contigs hold, again, from the records."""
        self._chk(self._lib.gd_rebuild_derived(self._ctx))

    def compute_launch(self):
        """gd_compute_launch: enqueue a compute, do not wait (compute_finish does)."""
        self._chk(self._lib.gd_compute_launch(self._ctx))

    def compute_finish(self):
        self._chk(self._lib.gd_compute_finish(self._ctx))

    def set_profiling(self, on: bool):
        self._chk(self._lib.gd_set_profiling(self._ctx, int(on)))

    def kernel_ms(self, kernel_id: int) -> float:
        ms = C.c_float()
        self._chk(self._lib.gd_kernel_ms(self._ctx, kernel_id, C.byref(ms)))
        return float(ms.value)

    def compute_timing(self):
        """Host wall clock of the last compute: dict(prepare_s, enqueue_s, wait_s, total_s) (gd_compute_timing)."""
        t = (C.c_double * 4)()
        self._chk(self._lib.gd_compute_timing(self._ctx, t, 4))
        return {"prepare_s": t[0], "enqueue_s": t[1], "wait_s": t[2], "total_s": t[3]}

    def stats(self) -> GdStats:
        s = GdStats()
        self._chk(self._lib.gd_get_stats(self._ctx, C.byref(s)))
        return s

    # -- results ----------------------------------------------------------
    def perbase(self, tid: int, start: int = 0, end: Optional[int] = None) -> np.ndarray:
        if end is None:
            end = self.contig_lengths[tid]
        out = np.empty(max(0, end - start), np.int32)
        self._chk(self._lib.gd_perbase(self._ctx, tid, start, end, out.ctypes.data))
        return out

    def window_sums(self, tid: int) -> np.ndarray:
        """int64 window sums only (works in every output mode)."""
        n = C.c_size_t()
        rc = self._lib.gd_windows(self._ctx, tid, None, None, 0, C.byref(n))
        if rc not in (0, -8):
            self._chk(rc)
        sums = np.zeros(n.value, np.int64)
        if n.value:
            self._chk(self._lib.gd_windows(self._ctx, tid, sums.ctypes.data, None, n.value, C.byref(n)))
        return sums

    def windows(self, tid: int):
        n = C.c_size_t()
        W = self.params.window_size
        cap = (self.contig_lengths[tid] + W - 1) // W
        sums = np.empty(cap, np.int64)
        mins = np.empty(cap, np.int32)
        self._chk(self._lib.gd_windows(self._ctx, tid, sums.ctypes.data, mins.ctypes.data, cap,
                                       C.byref(n)))
        return sums[:n.value], mins[:n.value]

    def _runs(self, fn, *args):
        n = C.c_size_t()
        cap = 1024
        while True:
            buf = (GdRun * cap)()
            rc = fn(self._ctx, *args, buf, cap, C.byref(n))
            if rc == -8:  # GD_E_CAPACITY
                cap = n.value
                continue
            self._chk(rc)
            a = np.frombuffer(buf, dtype=np.int32, count=3 * n.value).reshape(-1, 3).copy()
            return a

    def callable_runs(self, tid: int) -> np.ndarray:
        """[n,3] int32 rows (start, end, class)."""
        return self._runs(self._lib.gd_callable, tid)

    def region_windows(self, tid: int, start: int, end: int):
        n = C.c_size_t()
        W = self.params.window_size
        cap = max(1, (end - 1) // W - start // W + 1) if end > start else 1
        sums = np.empty(cap, np.int64)
        mins = np.empty(cap, np.int32)
        self._chk(self._lib.gd_region_windows(self._ctx, tid, start, end, sums.ctypes.data,
                                              mins.ctypes.data, cap, C.byref(n)))
        return sums[:n.value], mins[:n.value]

    def region_callable(self, tid: int, start: int, end: int) -> np.ndarray:
        return self._runs(self._lib.gd_region_callable, tid, start, end)

    def regions(self, tids, starts, ends):
        """Many regions in one call (gd_regions): returns ([sums], [mins], [runs]) with one array per region."""
        t = np.ascontiguousarray(tids, np.int32)
        s = np.ascontiguousarray(starts, np.int64)
        e = np.ascontiguousarray(ends, np.int64)
        n = len(t)
        W = int(self.params.window_size)
        nw = int(sum((int(b) - 1) // W - int(a) // W + 1 for a, b in zip(s, e) if b > a))
        sums, mins = np.zeros(max(nw, 1), np.int64), np.zeros(max(nw, 1), np.int32)
        woff, roff = np.zeros(n + 1, np.uint64), np.zeros(n + 1, np.uint64)
        cap = max(1024, 4 * n)
        while True:
            runs = (GdRun * cap)()
            rc = self._lib.gd_regions(self._ctx, n, t.ctypes.data, s.ctypes.data, e.ctypes.data, sums.ctypes.data,
                                      mins.ctypes.data, nw, woff.ctypes.data, runs, cap, roff.ctypes.data)
            if rc == -8 and int(woff[n]) <= nw and int(roff[n]) > cap:      # GD_E_CAPACITY on the runs: retry
                cap = int(roff[n])
                continue
            self._chk(rc)
            break
        allruns = np.frombuffer(runs, dtype=np.int32, count=3 * int(roff[n])).reshape(-1, 3)
        out_s, out_m, out_r = [], [], []
        for k in range(n):
            a, b = int(woff[k]), int(woff[k + 1])
            out_s.append(sums[a:b].copy())
            out_m.append(mins[a:b].copy())
            out_r.append(allruns[int(roff[k]):int(roff[k + 1])].copy())
        return out_s, out_m, out_r

    def depthwed(self, tids, size: int):
        """Sites x samples matrix of `goleft depthwed -s size`.  tids: [n_samples][n_ctg]
        engine contigs.  Returns (cells int64 [rows, samples], row_ctg, row_start, row_end)."""
        t = np.ascontiguousarray(tids, np.int32)
        assert t.ndim == 2
        ns, nc = t.shape
        n = C.c_size_t()
        rc = self._lib.gd_depthwed(self._ctx, ns, nc, t.ctypes.data, size, None, None, None, None, 0, C.byref(n))
        if rc not in (0, -8):
            self._chk(rc)
        rows = n.value
        cells = np.empty((rows, ns), np.int64)
        ctg = np.empty(rows, np.int32)
        st = np.empty(rows, np.int64)
        en = np.empty(rows, np.int64)
        if rows:
            self._chk(self._lib.gd_depthwed(self._ctx, ns, nc, t.ctypes.data, size, cells.ctypes.data,
                                            ctg.ctypes.data, st.ctypes.data, en.ctypes.data, rows, C.byref(n)))
        return cells, ctg, st, en

    def depthwed_device(self, tids, size: int):
        """The depthwed matrix left in HBM: (device pointer to int64 [rows][samples], rows)."""
        t = np.ascontiguousarray(tids, np.int32)
        assert t.ndim == 2
        p, n = C.c_void_p(), C.c_size_t()
        self._chk(self._lib.gd_depthwed_device(self._ctx, t.shape[0], t.shape[1], t.ctypes.data, size,
                                               C.byref(p), C.byref(n)))
        return p.value, n.value

    def seq_load(self, seq) -> None:
        """One contig's reference bases (bytes / uint8 array, FASTA line breaks removed) into HBM."""
        a = np.frombuffer(seq, np.uint8) if isinstance(seq, (bytes, bytearray)) else np.ascontiguousarray(seq, np.uint8)
        self._chk(self._lib.gd_seq_load(self._ctx, a.ctypes.data if a.size else None, a.size))

    def seq_stats(self, starts, ends):
        """`--stats` counts per window [start, end): (n_gc, n_cpg, n_masked) uint32 arrays
        (depth/depth.go:191-200; fractions and %.3g are the host's)."""
        s = np.ascontiguousarray(starts, np.int64)
        e = np.ascontiguousarray(ends, np.int64)
        assert s.shape == e.shape and s.ndim == 1
        gc = np.empty(s.size, np.uint32)
        cpg = np.empty(s.size, np.uint32)
        low = np.empty(s.size, np.uint32)
        self._chk(self._lib.gd_seq_stats(self._ctx, s.size, s.ctypes.data, e.ctypes.data,
                                         gc.ctypes.data, cpg.ctypes.data, low.ctypes.data))
        return gc, cpg, low

    def seq_stats_ex(self, starts, ends, line_bases: int = 0):
        """gd_seq_stats_ex: (n_gc, n_cpg, n_masked, n_acgt, n_masked_acgt) per window; line_bases > 0: a C at the
        end of a FASTA line starts no CpG (include/goleft_depth_host.h GDH_STATS_CPG_RAW_LINES)."""
        s = np.ascontiguousarray(starts, np.int64)
        e = np.ascontiguousarray(ends, np.int64)
        assert s.shape == e.shape and s.ndim == 1
        out = [np.empty(s.size, np.uint32) for _ in range(5)]
        self._chk(self._lib.gd_seq_stats_ex(self._ctx, s.size, s.ctypes.data, e.ctypes.data, int(line_bases),
                                            *[a.ctypes.data for a in out]))
        return tuple(out)

    def md_flags(self, tids, min_cov: int, min_samples: int):
        """multidepth: (any, suf) boolean arrays over the samples' common contig
        (multidepth/multidepth.go:163-171; bit x of word x // 32 on the wire)."""
        t = np.ascontiguousarray(tids, np.int32)
        length = self.contig_lengths[int(t[0])]
        nw = (length + 31) // 32
        a = np.zeros(nw, np.uint32)
        s = np.zeros(nw, np.uint32)
        self._chk(self._lib.gd_md_flags(self._ctx, t.size, t.ctypes.data, min_cov, min_samples,
                                        a.ctypes.data, s.ctypes.data, nw))
        unpack = lambda w: np.unpackbits(w.view(np.uint8), bitorder="little")[:length].astype(bool)
        return unpack(a), unpack(s)

    def md_begin(self, length: int):
        """multidepth with the samples in groups: zero the per-position accumulators."""
        self._chk(self._lib.gd_md_begin(self._ctx, int(length)))
        self._md_len = int(length)

    def md_accumulate(self, tids, min_cov: int):
        t = np.ascontiguousarray(tids, np.int32)
        self._chk(self._lib.gd_md_accumulate(self._ctx, t.size, t.ctypes.data, int(min_cov)))

    def md_finish(self, min_samples: int):
        """-> (any, suf) boolean arrays; they become the bitmaps of md_blocks / md_sums_group."""
        length = self._md_len
        nw = (length + 31) // 32
        a = np.zeros(max(nw, 1), np.uint32)
        s = np.zeros(max(nw, 1), np.uint32)
        self._chk(self._lib.gd_md_finish(self._ctx, int(min_samples), a.ctypes.data, s.ctypes.data, nw))
        unpack = lambda w: np.unpackbits(w.view(np.uint8), bitorder="little")[:length].astype(bool)
        return unpack(a), unpack(s)

    def md_load_flags(self, any_mask, suf_mask):
        a = np.asarray(any_mask, bool)
        s = np.asarray(suf_mask, bool)
        n = a.size
        pack = lambda m: np.ascontiguousarray(np.packbits(np.concatenate([m, np.zeros((-n) % 32, bool)]),
                                                          bitorder="little").view(np.uint32))
        aw, sw = pack(a), pack(s)
        self._chk(self._lib.gd_md_load_flags(self._ctx, aw.ctypes.data if n else None, sw.ctypes.data if n else None, n))

    def md_blocks(self, chunk: int, max_skip: int = 10, min_size: int = 15, window: int = 10000000) -> np.ndarray:
        """The block state machine of multidepth.go:188-268 on the device over the current bitmaps:
        int64 [n_blocks, 2] of {start, end}."""
        n = C.c_size_t(0)
        rc = self._lib.gd_md_blocks(self._ctx, int(chunk), int(max_skip), int(min_size), int(window), None, None, 0, C.byref(n))
        if rc not in (0, -8):
            self._chk(rc)
        st = np.zeros(max(1, n.value), np.int64)
        en = np.zeros(max(1, n.value), np.int64)
        if n.value:
            self._chk(self._lib.gd_md_blocks(self._ctx, int(chunk), int(max_skip), int(min_size), int(window),
                                             st.ctypes.data, en.ctypes.data, n.value, C.byref(n)))
        return np.stack([st[:n.value], en[:n.value]], 1)

    def md_sums_group(self, tids, starts, ends) -> np.ndarray:
        t = np.ascontiguousarray(tids, np.int32)
        s = np.ascontiguousarray(starts, np.int64)
        e = np.ascontiguousarray(ends, np.int64)
        out = np.zeros((s.size, t.size), np.float64)
        self._chk(self._lib.gd_md_sums_group(self._ctx, t.size, t.ctypes.data, s.size, s.ctypes.data, e.ctypes.data,
                                             out.ctypes.data))
        return out

    def md_sums(self, starts, ends, n_samples: int) -> np.ndarray:
        """multidepth: float64 [n_blocks, n_samples] running sums of depth / 1000 over the suf
        sites of each block, in position order (multidepth.go:270-277)."""
        s = np.ascontiguousarray(starts, np.int64)
        e = np.ascontiguousarray(ends, np.int64)
        out = np.zeros((s.size, n_samples), np.float64)
        self._chk(self._lib.gd_md_sums(self._ctx, s.size, s.ctypes.data, e.ctypes.data, out.ctypes.data))
        return out

    def inflate_bgzf(self, data: bytes, check_crc: bool = True):
        """Inflate every member of a BGZF byte string on the device (CRC32 of each member
        verified unless check_crc is False); returns (bytes, status[n])."""
        raw = np.frombuffer(data, np.uint8)
        offs, lens, isz, crcs = [], [], [], []
        p, n = 0, raw.size
        while p + 18 <= n:
            assert raw[p] == 0x1f and raw[p + 1] == 0x8b and raw[p + 3] & 4, "not a BGZF member at %d" % p
            xlen = int(raw[p + 10]) | int(raw[p + 11]) << 8
            q, bsize = p + 12, None
            while q < p + 12 + xlen:
                slen = int(raw[q + 2]) | int(raw[q + 3]) << 8
                if raw[q] == 66 and raw[q + 1] == 67 and slen == 2:
                    bsize = int(raw[q + 4]) | int(raw[q + 5]) << 8
                q += 4 + slen
            assert bsize is not None
            offs.append(p + 12 + xlen)
            lens.append(bsize + 1 - 12 - xlen - 8)
            isz.append(int.from_bytes(raw[p + bsize - 3:p + bsize + 1].tobytes(), "little"))
            crcs.append(int.from_bytes(raw[p + bsize - 7:p + bsize - 3].tobytes(), "little"))
            p += bsize + 1
        m = len(offs)
        in_off = np.asarray(offs, np.uint64)
        in_len = np.asarray(lens, np.uint32)
        out_len = np.asarray(isz, np.uint32)
        crc = np.asarray(crcs, np.uint32)
        out_off = np.zeros(m, np.uint64)
        if m:
            out_off[1:] = np.cumsum(out_len[:-1].astype(np.uint64))
        total = int(out_len.astype(np.uint64).sum())
        out = np.empty(max(total, 1), np.uint8)
        status = np.zeros(max(m, 1), np.uint32)
        self._chk(self._lib.gd_inflate_bgzf(self._ctx, raw.ctypes.data, raw.size, m, in_off.ctypes.data,
                                            in_len.ctypes.data, out_off.ctypes.data, out_len.ctypes.data,
                                            crc.ctypes.data if check_crc else None,
                                            out.ctypes.data, total, status.ctypes.data))
        return out[:total].tobytes(), status[:m]

    def ingest_bgzf(self, tid: int, data: bytes, base_coffset: int, anchors, ref_id: int | None = None) -> int:
        """Inflate + decode the records of BAM reference ref_id (default: tid) on the device into
        engine contig tid (see gd_ingest_bgzf); returns the record count."""
        raw = np.frombuffer(data, np.uint8)
        a = np.ascontiguousarray(anchors, np.uint64)
        n = C.c_uint64()
        self._chk(self._lib.gd_ingest_bgzf(self._ctx, tid, tid if ref_id is None else ref_id,
                                           raw.ctypes.data, raw.size, base_coffset,
                                           a.ctypes.data, a.size, C.byref(n)))
        return int(n.value)

    def ingest_bgzf_stream(self, tid: int, data: bytes, base_coffset: int, anchors, piece: int, ref_id=None) -> int:
        """The same read as ingest_bgzf, fed in pieces of `piece` bytes (gd_ingest_begin / feed / finish)."""
        raw = np.frombuffer(data, np.uint8)
        nm = C.c_size_t()
        rc = self._lib.gd_bgzf_members(raw.ctypes.data, raw.size, 0, None, None, None, None, None, C.byref(nm))
        if rc not in (0, -8):
            self._chk(rc)
        n = nm.value
        moff, msize, mhdr = np.zeros(n, np.uint64), np.zeros(n, np.uint32), np.zeros(n, np.uint16)
        misz, mcrc = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        self._chk(self._lib.gd_bgzf_members(raw.ctypes.data, raw.size, n, moff.ctypes.data, msize.ctypes.data,
                                            mhdr.ctypes.data, misz.ctypes.data, mcrc.ctypes.data, C.byref(nm)))
        used = int(moff[-1]) + int(msize[-1])
        self._chk(self._lib.gd_ingest_begin(self._ctx, used, base_coffset, n, moff.ctypes.data, msize.ctypes.data,
                                            mhdr.ctypes.data, misz.ctypes.data, mcrc.ctypes.data))
        for off in range(0, used, piece):
            k = min(piece, used - off)
            self._chk(self._lib.gd_ingest_feed(self._ctx, raw[off:off + k].ctypes.data, k))
        a = np.ascontiguousarray(anchors, np.uint64)
        cnt = C.c_uint64()
        self._chk(self._lib.gd_ingest_finish(self._ctx, tid, tid if ref_id is None else ref_id, a.ctypes.data, a.size,
                                             C.byref(cnt)))
        return int(cnt.value)

    def ingest_feed_range(self, data: bytes, base_coffset: int, piece: int = 32 << 20) -> None:
        """gd_ingest_begin + gd_ingest_feed of one byte range (it then waits for ingest_decode)."""
        raw = np.frombuffer(data, np.uint8)
        nm = C.c_size_t()
        rc = self._lib.gd_bgzf_members(raw.ctypes.data, raw.size, 0, None, None, None, None, None, C.byref(nm))
        if rc not in (0, -8):
            self._chk(rc)
        n = nm.value
        moff, msize, mhdr = np.zeros(n, np.uint64), np.zeros(n, np.uint32), np.zeros(n, np.uint16)
        misz, mcrc = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        self._chk(self._lib.gd_bgzf_members(raw.ctypes.data, raw.size, n, moff.ctypes.data, msize.ctypes.data,
                                            mhdr.ctypes.data, misz.ctypes.data, mcrc.ctypes.data, C.byref(nm)))
        used = int(moff[-1]) + int(msize[-1])
        self._chk(self._lib.gd_ingest_begin(self._ctx, used, base_coffset, n, moff.ctypes.data, msize.ctypes.data,
                                            mhdr.ctypes.data, misz.ctypes.data, mcrc.ctypes.data))
        for off in range(0, used, piece):
            k = min(piece, used - off)
            self._chk(self._lib.gd_ingest_feed(self._ctx, raw[off:off + k].ctypes.data, k))

    def ingest_decode(self, tid: int, ref_id: int, anchors) -> int:
        """One reference of the OLDEST pending range -> contig tid (gd_ingest_decode)."""
        a = np.ascontiguousarray(anchors, np.uint64)
        cnt = C.c_uint64()
        self._chk(self._lib.gd_ingest_decode(self._ctx, tid, ref_id, a.ctypes.data, a.size, C.byref(cnt)))
        return int(cnt.value)

    def ingest_decode_part(self, tid: int, ref_id: int, anchors, end_anchor: int = 0, append: bool = False,
                           release: bool = False, expect_scale: float = 0.0) -> int:
        """One PART of a reference from the oldest pending range (gd_ingest_decode_part): its anchors up to
        end_anchor (0: the range's end), appended to / replacing the contig's records."""
        a = np.ascontiguousarray(anchors, np.uint64)
        cnt = C.c_uint64()
        self._chk(self._lib.gd_ingest_decode_part(self._ctx, tid, ref_id, a.ctypes.data, a.size, int(end_anchor),
                                                  (1 if append else 0) | (2 if release else 0), float(expect_scale), C.byref(cnt)))
        return int(cnt.value)

    def ingest_release(self) -> None:
        self._chk(self._lib.gd_ingest_release(self._ctx))

    def ingest_bgzf_refs(self, data: bytes, base_coffset: int, refs, piece: int = 32 << 20):
        """One fed byte range that holds several references: refs = [(tid, ref_id, anchors), ...] in
        file order.  Returns the record counts."""
        self.ingest_feed_range(data, base_coffset, piece)
        counts = [self.ingest_decode(tid, ref_id, anchors) for tid, ref_id, anchors in refs]
        self.ingest_release()
        return counts

    def device_windows(self):
        """(ptr_sums, ptr_mins, n_total) device views of the concatenated window arrays."""
        ps, pm, n = C.c_void_p(), C.c_void_p(), C.c_size_t()
        self._chk(self._lib.gd_device_windows(self._ctx, C.byref(ps), C.byref(pm), C.byref(n)))
        return ps.value, pm.value, n.value

    def device_runs(self):
        p, n = C.c_void_p(), C.c_size_t()
        self._chk(self._lib.gd_device_runs(self._ctx, C.byref(p), C.byref(n)))
        return p.value, n.value

    def device_perbase(self, tid: int):
        p, n = C.c_void_p(), C.c_int64()
        self._chk(self._lib.gd_device_perbase(self._ctx, tid, C.byref(p), C.byref(n)))
        return p.value, n.value

    def get_option(self, option: int) -> int:
        v = C.c_int64(0)
        self._chk(self._lib.gd_get_option(self._ctx, int(option), C.byref(v)))
        return int(v.value)

    def check_commits(self):
        """gd_check_commits: the verdict on the blocks committed under OPT_COMMIT_CHECK = 1 (raises GdError)."""
        self._chk(self._lib.gd_check_commits(self._ctx))

    def set_option(self, option: int, value: int):
        """gd_set_option (OPT_* above): tuning / diagnostic switches; results never change."""
        self._chk(self._lib.gd_set_option(self._ctx, int(option), int(value)))

    def set_export(self, device_ptr: int, max_windows: int, cap_bounds: int):
        """gd_set_export: every compute() also writes the packed block
        [n_bounds][sums][mins][bounds] into caller-owned device memory (0 switches it off)."""
        self._chk(self._lib.gd_set_export(self._ctx, C.c_void_p(device_ptr) if device_ptr else None,
                                          int(max_windows), int(cap_bounds)))

    def comm_init(self, rank: int, world: int, unique_id: bytes):
        """gd_comm_init: this context becomes rank `rank` of `world` (collective; RCCL opened on first use)."""
        assert len(unique_id) >= 128
        buf = (C.c_char * 128).from_buffer_copy(unique_id[:128])
        self._chk(self._lib.gd_comm_init(self._ctx, int(rank), int(world), buf, 128))

    def comm_destroy(self):
        self._chk(self._lib.gd_comm_destroy(self._ctx))

    def gather_export(self, recv_ptr: int = 0, words: int = 0, root: int = 0, send_ptr: int = 0):
        """gd_gather_export: the export block (or send_ptr) to `root`, asynchronous on the context's copy stream."""
        self._chk(self._lib.gd_gather_export(self._ctx, C.c_void_p(send_ptr) if send_ptr else None,
                                             C.c_void_p(recv_ptr) if recv_ptr else None, int(words), int(root)))

    def gather_wait(self):
        self._chk(self._lib.gd_gather_wait(self._ctx))

    def wait_event(self, hip_event: int):
        """gd_wait_event: the engine's stream waits for a hipEvent_t (e.g. torch.cuda.Event().cuda_event)."""
        self._chk(self._lib.gd_wait_event(self._ctx, C.c_void_p(int(hip_event))))

    def window_offset(self, tid: int):
        o, n = C.c_size_t(), C.c_size_t()
        self._chk(self._lib.gd_window_offset(self._ctx, tid, C.byref(o), C.byref(n)))
        return o.value, n.value


def device_count() -> int:
    n = C.c_int()
    _lib.load().gd_device_count(C.byref(n))
    return n.value


def comm_library():
    """gd_comm_library: (path of the RCCL the library's collective uses, whether the process had it mapped already -- torch's
    own copy) or None when there is none (GD_E_NODEVICE).  Local: no other rank takes part."""
    buf = C.create_string_buffer(1024)
    shared = C.c_int(0)
    rc = _lib.load().gd_comm_library(buf, 1024, C.byref(shared))
    if rc != 0:
        return None
    return buf.value.decode(), bool(shared.value)


def comm_unique_id() -> bytes:
    """gd_comm_unique_id: 128 bytes that every rank of a communicator must be given (RCCL's ncclUniqueId)."""
    buf = (C.c_char * 128)()
    rc = _lib.load().gd_comm_unique_id(buf, 128)
    if rc != 0:
        raise GdError(rc, "gd_comm_unique_id failed (RCCL not available?)")
    return bytes(buf.raw)
