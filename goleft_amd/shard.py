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


def gather_to_root(sums: torch.Tensor, mins: torch.Tensor, bounds: torch.Tensor,
                   assignment: List[List[int]], lengths: Sequence[int], W: int,
                   rank: int, world: int, group=None):
    """Gather per-rank results to rank 0.

    Every rank passes its concatenated window sums/mins (contigs in ascending
    tid order) and its ordered run boundaries {pos, cls | local_index << 2}.
    Returns the gathered packed buffers on rank 0 (see unpack_gathered, which
    yields tid -> {"sums": i64[n_win], "mins": i32[n_win], "bounds": i32[k,2]})
    and None on the other ranks.
    Two collectives: an all_gather of boundary counts (8 bytes per rank), then
    one gather of a packed, padded int64 buffer."""
    dev = sums.device
    nwin = [sum(n_windows(lengths[t], W) for t in assignment[r]) for r in range(world)]
    nb_local = torch.tensor([bounds.numel() // 2], dtype=torch.int64, device=dev)
    if world == 1:
        counts = [int(nb_local.item())]
    else:
        cl = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(cl, nb_local, group=group)
        counts = [int(c.item()) for c in cl]
    max_w, max_b = max(nwin), max(counts)
    # packed layout (int64 words): [sums max_w][mins max_w as i32 pairs][bounds max_b]
    words_m = (max_w + 1) // 2
    total = max_w + words_m + max_b
    buf = torch.zeros(total, dtype=torch.int64, device=dev)
    k = sums.numel()
    buf[:k] = sums
    buf[max_w:max_w + words_m].view(torch.int32)[:k] = mins
    buf[max_w + words_m:max_w + words_m + counts[rank]] = bounds.view(torch.int64) \
        if bounds.numel() else bounds.new_zeros(0, dtype=torch.int64)
    if world == 1:
        parts = [buf]
    elif rank == 0:
        parts = [torch.empty_like(buf) for _ in range(world)]
        dist.gather(buf, parts, dst=0, group=group)
    else:
        dist.gather(buf, None, dst=0, group=group)
        return None
    return {"parts": parts, "counts": counts, "max_w": max_w, "words_m": words_m,
            "assignment": assignment, "lengths": list(lengths), "W": W}


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
