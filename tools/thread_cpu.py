#!/usr/bin/env python3
"""CPU time per thread of a command, sampled from /proc while it runs (threads that exit keep their last sample).
    python tools/thread_cpu.py <command ...>      -> wall, then (ticks of 10 ms, thread name) sorted, threads grouped by name"""
import os
import subprocess
import sys
import time

p = subprocess.Popen(sys.argv[1:], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
seen = {}
t0 = time.perf_counter()
while p.poll() is None:
    try:
        for tid in os.listdir("/proc/%d/task" % p.pid):
            try:
                with open("/proc/%d/task/%s/stat" % (p.pid, tid)) as fh:
                    s = fh.read()
                name = s[s.index("(") + 1:s.rindex(")")]
                f = s[s.rindex(")") + 2:].split()
                seen[tid] = (int(f[11]), int(f[12]), name)
            except (OSError, ValueError):
                pass
    except OSError:
        break
    time.sleep(0.02)
wall = time.perf_counter() - t0
print("wall %.2f s, %d threads seen, user %.2f s + system %.2f s (last samples)" % (
    wall, len(seen), sum(v[0] for v in seen.values()) / 100, sum(v[1] for v in seen.values()) / 100))
for tid, (u, sy, name) in sorted(seen.items(), key=lambda kv: -(kv[1][0] + kv[1][1]))[:40]:
    print("  %6s %-16s user %5.2f sys %5.2f" % (tid, name, u / 100, sy / 100))
