"""Chromosome -> GPU sharding and the final gather.

The reference parallelises `goleft depth` over genome tiles with a process
pool on one host (/root/reference/depth/depth.go:392-394, `-p`); results meet
again in the merge loop (:394-421).  Here contigs are the shard unit: each rank
(one process per GPU) computes depth for its contigs with no data-path
collective, then window sums / minima and coverage-class run boundaries are
gathered to rank 0 (RCCL over xGMI when the backend is "nccl"; gloo in the CPU
tests), which owns the BED output exactly like the reference's merge loop.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import torch
import torch.distributed as dist


def lpt_assign(lengths: Sequence[int], n_ranks: int) -> List[List[int]]:
    """Longest-processing-time-first assignment of contigs to ranks.

    Deterministic (ties broken by tid).  Returns per-rank tid lists, each
    sorted ascending so a rank's result arrays are in genome order."""
    order = sorted(range(len(lengths)), key=lambda t: (-int(lengths[t]), t))
    load = [0] * n_ranks
    out: List[List[int]] = [[] for _ in range(n_ranks)]
    for t in order:
        r = min(range(n_ranks), key=lambda k: (load[k], k))
        out[r].append(t)
        load[r] += int(lengths[t])
    return [sorted(x) for x in out]


def n_windows(length: int, W: int) -> int:
    return (int(length) + W - 1) // W


class _DevArray:
    """Expose a raw device pointer through __cuda_array_interface__."""

    def __init__(self, ptr: int, n: int, typestr: str):
        self.__cuda_array_interface__ = {"data": (int(ptr), False), "shape": (int(n),),
                                         "typestr": typestr, "version": 2}


def device_view(ptr: int, n: int, dtype: torch.dtype, device) -> torch.Tensor:
    """Zero-copy torch view of engine-owned device memory."""
    if n == 0 or not ptr:
        return torch.empty(0, dtype=dtype, device=device)
    typestr = {torch.int64: "<i8", torch.int32: "<i4"}[dtype]
    return torch.as_tensor(_DevArray(ptr, n, typestr), device=device)


def local_results(eng, device):
    """(sums i64[nw], mins i32[nw], bounds i32[2*nb]) device views of one engine."""
    ps, pm, nw = eng.device_windows()
    pb, nb = eng.device_runs()
    return (device_view(ps, nw, torch.int64, device), device_view(pm, nw, torch.int32, device),
            device_view(pb, 2 * nb, torch.int32, device))


class RootGather:
    """The path's one exchange step (depth/depth.go:394-421: the merge loop owns the two BED files):
    every rank's window sums / minima and ordered run boundaries to rank 0.

    Built ONCE, outside any timed region: fixed-capacity send / receive buffers are allocated here
    and reused by every step.  A step is then a few device-side copies into the packed send buffer
    and ONE collective (`gather`; RCCL posts it as point-to-point sends to rank 0, each peer over its
    own xGMI link) -- no `.item()`, no host synchronisation, no allocation.  The boundary count
    travels inside the buffer (word 0), so no count exchange precedes the payload.

    Packed layout per rank, int64 words:  [n_bounds][sums: max_w][mins: ceil(max_w / 2)][bounds: cap_b]
    where a boundary is the engine's {int32 pos, int32 cls | local_contig_index << 2} pair.

    `cap_b` (boundaries per rank) is agreed once by `reserve()` (an all_reduce MAX of the counts of
    a first compute; data-dependent, like the reference's callable.bed row count).  A later step
    with more boundaries than the capacity is detected on rank 0 by `result()` (`overflow`), never
    silently truncated.

    The buffers are DOUBLE: the collective of step k is asynchronous and reads send buffer k % 2 while
    the compute of step k + 1 already fills the other one (`gd_set_export` is re-pointed every step), so
    the exchange of one step runs under the kernels of the next -- and nothing is ever written into a
    buffer a collective may still be reading: before buffer k % 2 is handed out again (step k + 2) the
    collective of step k is waited for (two steps later it has long finished).  A pipelined caller (bench.py)
    orders a step as  compute_finish(k - 1); flip(); compute_launch(k); post()  -- the collective's host-side
    cost falls under the kernels of step k too; step_exported() is flip() + post() for a synchronous compute."""

    def __init__(self, assignment: List[List[int]], lengths: Sequence[int], W: int, rank: int,
                 world: int, device, bounds_cap: int = 1 << 16, group=None, native: bool = False):
        """native: the collective is the library's own (gd_gather_export: grouped ncclSend / ncclRecv on the attached
        engine's copy stream, RCCL opened by gd_comm_init) instead of torch.distributed's -- what a cgo host would
        call.  The engine must have been given a communicator (DepthEngine.comm_init) and be attach()ed."""
        self.assignment, self.lengths, self.W = assignment, list(lengths), int(W)
        self.rank, self.world, self.device, self.group = rank, world, device, group
        self.native = bool(native)
        self.nwin = [sum(n_windows(lengths[t], W) for t in assignment[r]) for r in range(world)]
        self.max_w = max(self.nwin) if self.nwin else 0
        self.words_m = (self.max_w + 1) // 2
        self.cap_b = 0
        self.send = self.recv = None
        self._eng = None
        self._alloc(int(bounds_cap))

    def _alloc(self, cap_b: int):
        self.drain()
        self.cap_b = cap_b
        self.total = 1 + self.max_w + self.words_m + cap_b
        self._sends = [torch.zeros(self.total, dtype=torch.int64, device=self.device) for _ in range(2)]
        self._recvs, self._partss = [None, None], [None, None]
        if self.rank == 0:
            self._recvs = [torch.zeros(self.world, self.total, dtype=torch.int64, device=self.device) for _ in range(2)]
            self._partss = [list(r.unbind(0)) for r in self._recvs]   # views of the receive buffers
        self._works = [None, None]
        self._events = [None, None]
        self._cur = 0                                           # the buffer pair the NEXT step uses
        self._last = 0                                          # the pair the last step used
        self.send, self.recv = self._sends[0], self._recvs[0]

    def _wait(self, k: int, host: bool = True):
        """The collective that last used buffer pair k has finished: on the host (host=True), or only in stream
        order -- the attached engine's stream then waits for it through an event and the host goes on (what flip()
        needs: the next compute must not write into a buffer the collective of two steps ago may still read; with
        a rank's step at ~0.5 ms a blocking wait here was the kind of fixed cost that decides 6x or 7x at N = 8)."""
        w = self._works[k] if getattr(self, "_works", None) else None
        if w == "native":
            # the library keeps its own order: the compute stream waits for the gather before the last one
            # (gd_gather_export); only a host-side wait has anything to do
            if host:
                self._eng.gather_wait()
            self._works[k] = None
            return
        if w is not None:
            w.wait()                                            # NCCL: the current torch stream waits; gloo: the host does
            cuda = self.device is not None and torch.device(self.device).type == "cuda"
            if cuda and not host and self._eng is not None and hasattr(self._eng, "wait_event"):
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(self.device))
                self._eng.wait_event(ev.cuda_event)
                self._events[k] = ev                            # (kept alive until the pair is waited for again)
            elif cuda:
                torch.cuda.current_stream(self.device).synchronize()
            self._works[k] = None

    def drain(self):
        """Every collective issued so far has finished (host side)."""
        for k in (0, 1):
            self._wait(k)

    def flip(self):
        """The current send buffer is complete (the compute that filled it has FINISHED): hand out the other pair.
        Its collective (two steps ago) must be over before anything writes into it again."""
        self._last = self._cur
        self._cur ^= 1
        self._wait(self._cur, host=False)
        self.send, self.recv = self._sends[self._cur], self._recvs[self._last]
        if self._eng is not None:
            self._eng.set_export(self.send.data_ptr(), self.max_w, self.cap_b)

    def post(self):
        """ONE collective on the buffer pair flip() retired, asynchronous."""
        k = self._last
        buf = self._sends[k]
        if self.native:
            self._eng.gather_export(recv_ptr=self._recvs[k].data_ptr() if self.rank == 0 else 0, words=self.total, root=0,
                                    send_ptr=buf.data_ptr())
            self._works[k] = "native"
        elif self.world == 1:
            self._recvs[k][0].copy_(buf)
        elif self.rank == 0:
            self._works[k] = dist.gather(buf, self._partss[k], dst=0, group=self.group, async_op=True)
        else:
            self._works[k] = dist.gather(buf, None, dst=0, group=self.group, async_op=True)

    def _exchange(self):
        self.flip()
        self.post()

    def reserve(self, n_bounds: int, slack: float = 1.25):
        """Collective, setup time only: make every rank's boundary capacity cover the largest count
        any rank has (times `slack`).  Returns the agreed capacity."""
        t = torch.tensor([int(n_bounds)], dtype=torch.int64, device=self.device)
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        need = int(int(t.item()) * slack) + 1024
        if need > self.cap_b:
            self._alloc(need)
        return self.cap_b

    def step(self, sums: torch.Tensor, mins: torch.Tensor, bounds: torch.Tensor):
        """Enqueue pack + gather on the current stream.  `bounds` is the flat int32 view (2 per
        boundary); its length is host-known (gd_compute returns the count), so nothing here waits
        for the device."""
        k = sums.numel()
        nb = bounds.numel() // 2
        buf = self._sends[self._cur]
        buf[0:1].fill_(nb)
        buf[1:1 + k].copy_(sums)
        buf[1 + self.max_w:1 + self.max_w + self.words_m].view(torch.int32)[:k].copy_(mins)
        m = min(nb, self.cap_b)
        if m:
            o = 1 + self.max_w + self.words_m
            buf[o:o + m].copy_(bounds.view(torch.int64)[:m])
        self._exchange()

    def attach(self, eng):
        """Let the engine fill the send buffer itself: gd_set_export makes every gd_compute write the
        packed block (same layout) before its one synchronisation, so a step is compute + ONE
        collective with no pack launches at all.  Call again after reserve() re-allocated."""
        self._eng = eng
        eng.set_export(self._sends[self._cur].data_ptr(), self.max_w, self.cap_b)
        self._attached = id(self._sends)

    def step_exported(self):
        """The exchange after a compute() of an attached engine (the block is already in the current send
        buffer); the engine's next compute exports into the other one."""
        assert getattr(self, "_attached", None) == id(self._sends), "attach() after the last (re)allocation"
        self._exchange()

    def result(self):
        """Rank 0: what the last step gathered, as the dict unpack_gathered() takes (one host
        synchronisation, outside the exchange); other ranks: None."""
        self._wait(self._last)
        if self.rank != 0:
            return None
        recv = self._recvs[self._last]
        counts = [int(c) for c in recv[:, 0].tolist()]
        return {"parts": [p[1:] for p in recv.unbind(0)], "counts": [min(c, self.cap_b) for c in counts],
                "overflow": any(c > self.cap_b for c in counts), "true_counts": counts,
                "max_w": self.max_w, "words_m": self.words_m,
                "assignment": self.assignment, "lengths": self.lengths, "W": self.W}


def gather_to_root(sums: torch.Tensor, mins: torch.Tensor, bounds: torch.Tensor,
                   assignment: List[List[int]], lengths: Sequence[int], W: int,
                   rank: int, world: int, group=None):
    """One-shot form (setup + one step + result): see RootGather.  Returns the gathered packed
    buffers on rank 0 (unpack_gathered yields tid -> {"sums", "mins", "bounds"}), None elsewhere."""
    g = RootGather(assignment, lengths, W, rank, world, sums.device, bounds_cap=0, group=group)
    g.reserve(bounds.numel() // 2, slack=1.0)
    g.step(sums, mins, bounds)
    return g.result()


def unpack_gathered(g) -> Dict[int, dict]:
    """Split what gather_to_root returned on rank 0 into per-contig results
    (host-side bookkeeping; not part of the exchange)."""
    out: Dict[int, dict] = {}
    max_w, words_m, W = g["max_w"], g["words_m"], g["W"]
    for r, p in enumerate(g["parts"]):
        s_all = p[:max_w]
        m_all = p[max_w:max_w + words_m].view(torch.int32)
        b_all = p[max_w + words_m:max_w + words_m + g["counts"][r]].view(torch.int32).view(-1, 2)
        local_idx = b_all[:, 1] >> 2
        off = 0
        for j, t in enumerate(g["assignment"][r]):
            nw = n_windows(g["lengths"][t], W)
            sel = b_all[local_idx == j]
            out[t] = {"sums": s_all[off:off + nw], "mins": m_all[off:off + nw],
                      "bounds": torch.stack([sel[:, 0], sel[:, 1] & 3], 1)}
            off += nw
    return out
