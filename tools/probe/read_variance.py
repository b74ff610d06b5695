#!/usr/bin/env python3
"""Why does the device BAM read's `pread` phase take 0.65 s per genome in one run and 2.05 s in the next (DESIGN.md
section 4, scope iii)?  One BAM (chr1-chr4 of the synthetic genome, 13 GB, in /dev/shm), the CLI run over it under a list
of conditions, interleaved with the default so that drift over time shows; per run: wall, the library's own phases, the
child's user / system CPU seconds, and what the container's CPU quota did meanwhile (cgroup cpu.stat).
    python tools/probe/read_variance.py [--reps 2] > gpurun_out/read_variance.jsonl     (GPU box; needs no torch)"""
import argparse
import ctypes as C
import json
import os
import resource
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def read(path):
    try:
        with open(path) as fh:
            return fh.read().strip()
    except OSError:
        return None


def cpu_stat():
    out = {}
    for ln in (read("/sys/fs/cgroup/cpu.stat") or "").splitlines():
        k, v = ln.split()
        out[k] = int(v)
    return out


def node_shmem():
    out = {}
    base = "/sys/devices/system/node"
    for n in sorted(x for x in os.listdir(base) if x.startswith("node")):
        for ln in (read("%s/%s/meminfo" % (base, n)) or "").splitlines():
            f = ln.split()
            if f[2] in ("Shmem:", "FilePages:", "MemFree:"):
                out["%s.%s" % (n, f[2][:-1])] = int(f[3]) // 1024          # MB
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--contigs", type=int, default=4)
    ap.add_argument("--pause-test", action="store_true", help="only: runs back to back, then with 1 / 2 / 4 s between them")
    args = ap.parse_args()
    from goleft_amd import synth
    lengths = list(synth.HG19_LENGTHS[:args.contigs])
    print(json.dumps({"host": {"nproc": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)), "cpu.max": read("/sys/fs/cgroup/cpu.max"),
                               "nodes": read("/sys/devices/system/node/online"),
                               "node0_cpus": read("/sys/devices/system/node/node0/cpulist"),
                               "node1_cpus": read("/sys/devices/system/node/node1/cpulist"),
                               "gpu_numa": [read(os.path.join("/sys/class/drm", c, "device/numa_node"))
                                            for c in sorted(os.listdir("/sys/class/drm")) if c.startswith("card") and "-" not in c][:4],
                               "thp_shmem": read("/sys/kernel/mm/transparent_hugepage/shmem_enabled"),
                               "thp": read("/sys/kernel/mm/transparent_hugepage/enabled"),
                               "before_write": node_shmem()}}), flush=True)
    d = tempfile.mkdtemp(prefix="gd_readvar_", dir="/dev/shm")
    try:
        bam = os.path.join(d, "synth.bam")
        t0 = time.perf_counter()
        info = json.loads(subprocess.check_output([os.path.join(ROOT, "goleft_amd", "synth-bam"), bam, "chrS",
                                                   ",".join(str(x) for x in lengths), "30", "20"]).decode())
        print(json.dumps({"bam_bytes": info["bam_bytes"], "write_s": time.perf_counter() - t0, "after_write": node_shmem()}), flush=True)
        exe = os.path.join(ROOT, "goleft_amd", "goleft-depth")
        node0 = read("/sys/devices/system/node/node0/cpulist")
        node1 = read("/sys/devices/system/node/node1/cpulist")

        def run(tag, env=None, prefix=(), pause=0.0):
            for rep in range(args.reps):
                time.sleep(pause)
                s0, r0, t0 = cpu_stat(), resource.getrusage(resource.RUSAGE_CHILDREN), time.perf_counter()
                p = subprocess.run(list(prefix) + [exe, "depth", "-w", "1000", "-p", "0", "-r", os.path.join(d, "synth.fa"), "--prefix",
                                                   os.path.join(d, "out"), bam],
                                   env=dict(os.environ, GOLEFT_DEPTH_TIMING="1", GOLEFT_INGEST_TIMING="1", **(env or {})),
                                   stderr=subprocess.PIPE, stdout=subprocess.DEVNULL)
                dt = time.perf_counter() - t0
                r1, s1 = resource.getrusage(resource.RUSAGE_CHILDREN), cpu_stat()
                ph = {}
                for ln in p.stderr.decode().strip().splitlines():
                    if ln.startswith("{"):
                        ph.update(json.loads(ln))
                print(json.dumps({"tag": tag, "rep": rep, "rc": p.returncode, "wall_s": round(dt, 3),
                                  "user_s": round(r1.ru_utime - r0.ru_utime, 2), "sys_s": round(r1.ru_stime - r0.ru_stime, 2),
                                  "throttled_periods": s1.get("nr_throttled", 0) - s0.get("nr_throttled", 0),
                                  "throttled_ms": (s1.get("throttled_usec", 0) - s0.get("throttled_usec", 0)) // 1000,
                                  "periods": s1.get("nr_periods", 0) - s0.get("nr_periods", 0),
                                  **{k: ph.get(k) for k in ("lib_read_s", "lib_wait_link_s", "lib_begin_s", "lib_wait_inflate_s",
                                                            "lib_count_walk_s", "lib_alloc_and_extract_walk_s", "setup_s", "read_s",
                                                            "listing_thread_s")}}), flush=True)

        if args.pause_test:
            run("settle")
            for pz in (0.0, 1.0, 2.0, 4.0, 0.0):
                run("pause_%.0fs" % pz, pause=pz)
            return
        run("default")
        for n in (4, 8, 12, 24):
            run("push_threads_%d" % n, {"GOLEFT_PUSH_THREADS": str(n)})
        run("default_again")
        if node0:
            run("taskset_node0", prefix=("taskset", "-c", node0))
        if node1:
            run("taskset_node1", prefix=("taskset", "-c", node1))
        run("taskset_16cpus", prefix=("taskset", "-c", "0-15"))
        run("dma_engine", {"GOLEFT_INGEST_DMA": "1"})
        run("mmap", {"GOLEFT_INGEST_MMAP": "1"})
        # a parent that holds a device context and 40 GB of HBM, as bench.py's process does while the CLI runs
        try:
            lib = C.CDLL(os.path.join(ROOT, "goleft_amd", "libgoleft_depth.so"))
            ctx, mem = C.c_void_p(), C.c_void_p()
            lib.gd_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
            lib.gd_device_alloc.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
            rc = lib.gd_create(0, C.byref(ctx))
            rc2 = lib.gd_device_alloc(ctx, 40 << 30, C.byref(mem)) if rc == 0 else None
            print(json.dumps({"parent_context": rc, "parent_alloc": rc2}), flush=True)
            run("parent_holds_context")
        except Exception as e:
            print(json.dumps({"parent_context_error": str(e)}), flush=True)
        run("default_last")
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
