#!/usr/bin/env python3
"""Measured HBM ceilings on this box (what the 8 TB/s spec peak means for a kernel whose traffic is
mostly writes): fill (write only), copy (read + write), sum (read only).  torch kernels, HIP events."""
import json
import torch

def timed(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n

def main():
    dev = torch.device("cuda:0")
    n = 3095677412                       # int32 elements: the per-base vector of the WGS workload
    x = torch.empty(n, dtype=torch.int32, device=dev)
    out = {"bytes": n * 4}
    ms = timed(lambda: x.fill_(1))
    out["fill_ms"], out["fill_TBps"] = ms, n * 4 / ms / 1e9
    ms = timed(lambda: x.zero_())
    out["memset_ms"], out["memset_TBps"] = ms, n * 4 / ms / 1e9
    h = n // 2
    ms = timed(lambda: x[h:2 * h].copy_(x[:h]))
    out["copy_ms"], out["copy_TBps_read_plus_write"] = ms, 2 * h * 4 / ms / 1e9
    ms = timed(lambda: x.sum())
    out["sum_ms"], out["read_TBps"] = ms, n * 4 / ms / 1e9
    print(json.dumps(out))

if __name__ == "__main__":
    main()
