#!/usr/bin/env python3
"""SURVEY.md section 8d scope (iii): BAM file -> BED, end to end, through the C++ host
(`goleft-depth depth`): BGZF inflate + record decode on the host cores, pinned ring, H2D,
kernels, BED rows.  Host-bound by construction; reported in DESIGN.md, never as bench `value`.

    python tools/scope3.py [--length 63025520] [--coverage 30] [--threads 0]
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--length", default="63025520", help="contig length, or several separated by commas")
    ap.add_argument("--coverage", type=float, default=30.0)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--window", type=int, default=1000)
    ap.add_argument("--name", default="chr20", help="contig name in the synthetic BAM")
    ap.add_argument("--paper", action="store_true",
                    help="the one invocation the reference itself times (indexcov/paper/cmp.sh:6): "
                         "`goleft depth --chrom <name> -p 20 -o -w 16384`")
    ap.add_argument("--variants", default="", help="extra device-decoder runs, each a comma list of ENV=VALUE, separated by ';' "
                                                   "(e.g. 'GOLEFT_INGEST_GROUP_MB=4096;GOLEFT_COPY_THREADS=16')")
    ap.add_argument("--no-host", action="store_true", help="skip the host-decoder run")
    ap.add_argument("--dir", default="", help="where the BAM is written (default: /dev/shm when it has the room, else /tmp)")
    ap.add_argument("--host-reps", type=int, default=3, help="repetitions of the host-decoder run (device runs: 3)")
    ap.add_argument("--pause", type=float, default=2.0,
                    help="seconds between the previous run's exit and the next start: a process that starts while the driver still "
                         "clears the device memory its predecessor released waits for it in its first large hipMalloc "
                         "(profiles/r11i_pause_test.jsonl: 2.5-3.4 s per run back to back, 1.06-1.26 s with >= 1 s between)")
    ap.add_argument("--rocprof", default="", help="directory: one more device-decoder run under rocprofv3 --kernel-trace --stats")
    args = ap.parse_args()
    import shutil
    need = sum(int(x) for x in args.length.split(",")) * 17
    # a RAM-backed directory when it has the room (the GPU boxes' /tmp is an overlay that writes at 0.3 GB/s)
    base = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > need * 1.3 else "/tmp"
    d = tempfile.mkdtemp(prefix="gd_scope3_", dir=args.dir or base)
    bam = os.path.join(d, "synth.bam")
    t0 = time.perf_counter()
    if args.paper:
        args.window, args.threads = 16384, 20
    info = json.loads(subprocess.check_output([os.path.join(ROOT, "goleft_amd", "synth-bam"), bam, args.name,
                                               args.length, str(args.coverage), "20"]).decode())
    subprocess.run(["sync"])                                # the file at rest: no write-back under the timed runs
    # one untimed pass over the file: the first read after a write moves the file's pages to the page cache's active list
    # under one lock (3x the wall time of every later read, profiles/r11g_read_variance.jsonl)
    with open(bam, "rb", buffering=0) as fh:
        buf = bytearray(64 << 20)
        while fh.readinto(buf):
            pass
    args.length = sum(int(x) for x in args.length.split(","))
    t_write = time.perf_counter() - t0
    extra = ["--chrom", args.name, "-o"] if args.paper else []
    out = {"scope": "BAM file -> depth.bed + callable.bed (goleft-depth CLI, process start to exit)",
           "invocation": "goleft-depth depth -w %d -p %d %s--prefix OUT synth.bam" % (args.window, args.threads, " ".join(extra) + (" " if extra else "")),
           "ref_bases": args.length, "coverage": args.coverage, "reads": info["reads"],
           "bam_MB": info["bam_bytes"] / 1e6, "host_cores": os.cpu_count(), "bam_write_s": t_write, "dir": d}
    beds = {}
    variants = [("device", {}), ("host", {"GOLEFT_GPU_DECODE": "0"})]
    if args.no_host:
        variants = variants[:1]
    for v in filter(None, args.variants.split(";")):
        variants.insert(1, ("device", dict(kv.split("=", 1) for kv in v.split(","))))
    for vi, (decoder, env) in enumerate(variants):
        best = None
        runs = []
        for rep in range(args.host_reps if decoder == "host" else 3):   # the file is in the page cache after the write
            time.sleep(args.pause)
            t0 = time.perf_counter()
            p = subprocess.run([os.path.join(ROOT, "goleft_amd", "goleft-depth"), "depth", "-w", str(args.window),
                                "-p", str(args.threads)] + extra + ["-r", os.path.join(d, "synth.fa"), "--prefix",
                                os.path.join(d, "out_" + decoder), bam],
                               env=dict(os.environ, GOLEFT_DEPTH_TIMING="1", GOLEFT_INGEST_TIMING="1", **env), stderr=subprocess.PIPE)
            dt = time.perf_counter() - t0
            assert p.returncode == 0, p.stderr.decode()
            if os.environ.get("SCOPE3_KEEP_STDERR"):       # (measurement builds print marks of their own)
                with open(os.environ["SCOPE3_KEEP_STDERR"], "a") as f:
                    f.write("== %s %s run %d\n%s\n" % (decoder, env, rep, p.stderr.decode()))
            lines = p.stderr.decode().strip().splitlines()
            phases = json.loads(lines[-1])
            assert phases["decoder"] == decoder, phases
            for ln in lines[:-1]:                          # GOLEFT_INGEST_TIMING: the device read's own phases
                if ln.startswith("{"):
                    phases.update(json.loads(ln))
            runs.append({"wall_s": dt, "lib_begin_s": phases.get("lib_begin_s"), "read_s": phases.get("read_s"), "setup_s": phases.get("setup_s")})
            if best is None or dt < best[0]:
                best = (dt, phases)
        dt, phases = best
        stem = os.path.join(d, "out_" + decoder) + ((".%s" % args.name) if args.paper else "")   # depth/depth.go:382-388
        beds[decoder] = open(stem + ".depth.bed").read() + open(stem + ".callable.bed").read()
        out[decoder + "_decoder" + (("_%d" % vi) if vi and decoder == "device" else "")] = {"env": env, "wall_s": dt, "ref_bases_per_s": args.length / dt,
                                     "bam_MB_per_s": info["bam_bytes"] / 1e6 / dt, "phases": phases, "all_runs": runs}
    out["outputs_identical"] = beds["device"] == beds["host"] if "host" in beds else None
    out["depth_rows"] = beds["device"].count("\n")
    if args.rocprof:
        subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "-d", args.rocprof, "-o", "scope3", "--output-format", "csv", "--",
                        os.path.join(ROOT, "goleft_amd", "goleft-depth"), "depth", "-w", str(args.window), "-p", str(args.threads)] + extra +
                       ["-r", os.path.join(d, "synth.fa"), "--prefix", os.path.join(d, "out_prof"), bam],
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=dict(os.environ, TMPDIR="/tmp", GOLEFT_SLOW_EXIT="1"))
    print(json.dumps(out))
    for f in os.listdir(d):
        os.unlink(os.path.join(d, f))
    os.rmdir(d)


if __name__ == "__main__":
    main()
