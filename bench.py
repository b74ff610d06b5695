#!/usr/bin/env python3
"""bench.py -- headline benchmark of the per-base depth engine.

Metric (BASELINE.json): reference bases per second of bit-exact per-base depth
(+ 1 kb window reduction + coverage-class runs) on a synthetic 30x WGS-shaped
decoded-record stream, 150 bp reads, hg19 contig lengths (3 095 677 412 bp).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload wgs|chr20]

A step = one pass of the hot path over the rank's HBM-resident record streams AS
THEY CROSSED THE C ABI (pos / flag / MAPQ / CSR offsets / BAM-encoded CIGAR ops,
nothing derived): gd_compute (prep + tile + run-ordering kernels, synchronous)
-- the short-read tile kernel reads those records directly -- and, where the
path needs derived structures (long reads: deletion lists, read records, tile
indexes), their construction (gd_rebuild_derived) INSIDE every step: what one `goleft depth`
run pays per input.
When ONE genome is shared by N > 1 GPUs a step also holds the gather of window
sums/minima and run boundaries to rank 0 over RCCL (the only exchange the path
has):

  --scaling strong (default) BASELINE.json config 3: N GPUs share ONE 3.1 Gb
                   genome, contigs assigned by LPT (goleft_amd/shard.py), every
                   step ends with the gather to rank 0 INSIDE the timed region
                   (one collective on a pre-allocated packed buffer the engine
                   fills itself, gd_set_export); `value` = 3.1 Gb / step.  With
                   N > 1 the line also carries the cohort case under
                   "cohort_weak_scaling" and the compute / gather split;
  --scaling weak   N GPUs process a cohort of N 30x genomes, one whole genome
                   per GPU (by sample): every rank owns its sample's outputs,
                   as N independent `goleft depth` runs would, so no exchange
                   step exists and none is timed.

Inputs are resident in HBM before the timed region starts.  Rank 0 prints one
JSON line.
"""
import argparse
import json
import os
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
HBM_ACHIEVABLE_GBPS = 6290.0   # MI355X_MICROARCH.md (HBM): what a streaming kernel reaches of the 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--samples", type=int, default=200, help="cohort workload: number of chr1 samples")
    ap.add_argument("--wed-size", type=int, default=1000, help="cohort workload: depthwed -s")
    ap.add_argument("--cohort-outputs", default="sums", choices=["sums", "windows"],
                    help="cohort workload: window sums only (default; all depthwed needs) or sums + minima + class runs")
    ap.add_argument("--workload", default="wgs", choices=["wgs", "chr20", "ont", "ont-chr20", "cohort"],
                    help="wgs/chr20: 30x 150 bp short reads (headline); ont/ont-chr20: 20x long reads "
                         "(BASELINE.json config 5, chunk path)")
    ap.add_argument("--coverage", type=float, default=None)
    ap.add_argument("--window", type=int, default=1000)
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-contigs", type=int, default=8)
    ap.add_argument("--no-host-stream", action="store_true",
                    help="skip the PCIe-inclusive scope (host records -> pinned ring -> results on host)")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE",
                    help="experiments only: gd_set_option(KEY, VALUE) on the engine (include/goleft_depth.h GD_OPT_*)")
    ap.add_argument("--emulate-shards", default="2,4,8", metavar="N[,N...]",
                    help="N = 1 only: time every LPT shard of the N-GPU job on THIS device, one after the other "
                         "(gd_select_contigs per shard, export block attached) -> per-shard ms and the projected "
                         "speed-up step(1) / max shard; '' = off")
    ap.add_argument("--bam-scope", default="auto", choices=["auto", "genome", "chr1-2", "chr20-21", "off"],
                    help="N = 1, wgs workload: BAM file -> BED through the CLI (SURVEY 8d scope iii) on a synthetic BAM of this "
                         "size; auto = genome on a host with >= 64 cores and the room for a 46 GB file, else chr20-21")
    ap.add_argument("--other-workloads", default="chr20,ont,cohort", metavar="W[,W...]",
                    help="N = 1, wgs workload: after the headline, short runs of BASELINE.json's configs 2 (chr20), 5 (ont) and "
                         "4 (cohort) -> `other_workloads` in the line; '' = off")
    ap.add_argument("--allow-gather-fallback", action="store_true",
                    help="N > 1: when the library's own collective (gd_gather_export) cannot be used, time torch.distributed's gather instead "
                         "and exit 0; without this flag such a run still prints its line and then exits with status 3 -- a fallback is never silent")
    ap.add_argument("--verify", action="store_true",
                    help="check one contig against the CPU oracle after timing")
    return ap.parse_args()


def usable_cpus():
    """The CPUs this process can really keep busy: os.cpu_count(), or fewer under a container's CPU quota (cgroup v2
    cpu.max / v1 cfs_quota_us).  The GPU boxes show 256 CPUs and grant 16 CPUs' worth of time (cpu.max 1600000 100000)."""
    n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, per = (fh.read().split() + ["100000"])[:2]
        if q != "max":
            n = max(1, min(n, -(-int(q) // int(per))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fh:
                q = int(fh.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
                per = int(fh.read())
            if q > 0 and per > 0:
                n = max(1, min(n, -(-q // per)))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(sample, W, mincov, cores):
    """Time the CPU oracle ("port": oracle/depth_oracle.c) on `sample` =
    [(name, length, Reads)], tiled in W-aligned 10 Mb regions like
    depth/depth.go:150-154, `cores` tiles in flight like `goleft depth -p`."""
    from oracle import pyoracle as po
    po.lib()
    jobs = []
    for name, length, r in sample:
        for s, e in po.tiles_for(length, W):
            lo = int(np.searchsorted(r.pos, max(0, s - 4096), "left"))
            hi = int(np.searchsorted(r.pos, e, "left"))
            jobs.append((name, s, e, r, lo, hi))
    td = tempfile.mkdtemp(prefix="gd_cpu_")

    def one(i):
        name, s, e, r, lo, hi = jobs[i]
        sub = r.slice(lo, hi)
        d = po.perbase_c(sub, 1, s, e, diff=True)
        hd = os.path.join(td, "%d.depth.bed" % i)
        ca = os.path.join(td, "%d.callable.bed" % i)
        po.callback_c(name, s, e, d, W, mincov, 0, hd, ca)
        return e - s

    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        bases = sum(ex.map(one, range(len(jobs))))
    dt = time.perf_counter() - t0
    for f in os.listdir(td):
        os.unlink(os.path.join(td, f))
    os.rmdir(td)
    return bases / dt, bases, dt


def real_samtools():
    """A samtools on PATH (or $SAMTOOLS) that is NOT this repository's samtools-shaped shim; None in the build image and on
    the GPU boxes of this pool ("if a samtools binary is ever found ... additionally time the true pipeline")."""
    import shutil
    import subprocess
    st = shutil.which(os.environ.get("SAMTOOLS", "samtools"))
    if not st or os.path.realpath(st).startswith(os.path.realpath(ROOT) + os.sep):
        return None
    try:
        p = subprocess.run([st, "--version"], capture_output=True, text=True, timeout=20)
    except (OSError, subprocess.SubprocessError):
        return None
    return (st, p.stdout.splitlines()[0]) if p.returncode == 0 and p.stdout else None


def reference_children(samtools, bam, contigs, W, Q, cores, max_tiles=64):
    """The reference's own child processes, timed: for the W-aligned 10 Mb tiles `goleft depth` makes (depth/depth.go:122-159)
    exactly the command it binds (`samtools depth -Q q -d 2500 -r chr:s-e bam`, depth/depth.go:45), `cores` of them in flight
    like its process.Runner, their text read and thrown away.  That is the reference pipeline WITHOUT its Go callback (the
    per-base parse, window means, class runs) -- an upper bound of what it delivers on this box's cores, on a bounded sample
    (`max_tiles` tiles)."""
    import subprocess
    from oracle import pyoracle as po
    jobs = [(name, s, e) for name, length in contigs for s, e in po.tiles_for(length, W)][:max_tiles]

    def one(j):
        name, s, e = j
        p = subprocess.Popen([samtools, "depth", "-Q", str(Q), "-d", "2500", "-r", "%s:%d-%d" % (name, s + 1, e), bam], stdout=subprocess.PIPE)
        n = 0
        for blk in iter(lambda: p.stdout.read(1 << 20), b""):
            n += len(blk)
        if p.wait() != 0:
            raise RuntimeError("samtools depth exited with %d" % p.returncode)
        return e - s, n

    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        res = list(ex.map(one, jobs))
    dt = time.perf_counter() - t0
    bases = sum(b for b, _ in res)
    return {"value": bases / dt, "unit": "ref-bases/s", "cores": cores, "kind": "reference",
            "what": "the reference's samtools children alone (depth/depth.go:45), %d in flight, text discarded: an upper bound of the "
                    "reference pipeline, which adds the Go callback's per-base parse (depth/depth.go:282-325)" % cores,
            "sample": "%d tiles of <= 10 Mb (%d ref-bases) of the bam_file_scope file" % (len(jobs), bases), "seconds": dt,
            "text_bytes": sum(n for _, n in res)}


def host_stream_scope(local_rank, W, Q, mincov, genome=False, reps=3, opts=()):
    """SURVEY.md section 8d scope (ii), reported next to -- never as -- `value`: the records of
    BASELINE.json config 2 (chr20; genome=True: config 3, the whole 30x genome) start in ordinary
    HOST memory, go through the library's pinned ring and over PCIe, are computed, and the window
    sums / minima and class runs come back to the host.  Everything a cgo caller would pay except
    the BAM decode itself.  Two ways in:
      push      gd_push from pageable arrays (the library's threads copy into the pinned blocks);
      in_place  gd_acquire -> the caller's threads write the block in place -> gd_commit, what
                INTEGRATION.md tells the Go host to do (its BAM decoder writes into the block; here
                the "decoder" is a pool of threads copying from the same arrays)."""
    from goleft_amd import synth
    from goleft_amd.engine import DepthEngine
    import torch
    dev = torch.device("cuda", local_rank)
    lengths = list(synth.HG19_LENGTHS) if genome else [synth.CHR20_LEN]
    seeds = list(range(1, len(lengths) + 1)) if genome else [20]
    recs = []
    for L, sd in zip(lengths, seeds):
        # generated on the device (the numpy twin is bit identical but slow), then moved to host memory
        t = [x.cpu().numpy() for x in synth.short_reads_torch(L, synth.n_reads_for(L), sd, dev)]
        recs.append((t[0], t[1].view(np.uint16), t[2], t[3].view(np.uint32), t[4].view(np.uint32)))
    torch.cuda.empty_cache()
    nbytes = sum(int(a.nbytes) for rec in recs for a in rec)
    n_reads = sum(int(rec[0].shape[0]) for rec in recs)
    fillers = int(os.environ.get("GOLEFT_BENCH_FILLERS", "0")) or min(16, max(2, (os.cpu_count() or 4) // 4))
    from goleft_amd import _hostlib
    hostlib = _hostlib.load()

    def in_place(eng):
        # the host library's producer: gd_acquire -> `fillers` threads write the block in place -> gd_commit
        for tid, (pos, flag, mapq, off, cig) in enumerate(recs):
            rc = hostlib.gdh_produce_in_place(eng._ctx, tid, pos.ctypes.data, flag.ctypes.data, mapq.ctypes.data,
                                              off.ctypes.data, cig.ctypes.data, pos.shape[0], cig.shape[0], fillers, 1 << 20)
            if rc != 0:
                raise RuntimeError("gdh_produce_in_place: %d" % rc)

    def push(eng):
        for tid, rec in enumerate(recs):
            eng.push(tid, *rec)

    out = {}
    with DepthEngine(local_rank) as eng:
        eng.set_params(window_size=W, min_mapq=Q, min_cov=mincov)
        for kv in opts:
            k, v = kv.split("=")
            eng.set_option(int(k), int(v))
        eng.set_contigs(lengths)
        for name, feed in (("push", push), ("in_place", in_place)):
            best = None
            for rep in range(reps + 1):                  # the first pass grows the device arrays and the ring (untimed)
                eng.reset()
                t0 = time.perf_counter()
                feed(eng)
                t1 = time.perf_counter()
                eng.compute()
                for tid in range(len(lengths)):
                    eng.windows(tid)
                    eng.callable_runs(tid)
                dt = time.perf_counter() - t0
                if rep > 0 and (best is None or dt < best[0]):
                    best = (dt, t1 - t0)
            out[name] = {"value": sum(lengths) / best[0], "unit": "ref-bases/s", "ms": best[0] * 1e3,
                         "feed_ms": best[1] * 1e3, "host_to_device_GBps": nbytes / best[0] / 1e9,
                         "feed_GBps": nbytes / best[1] / 1e9}
    better = max(out, key=lambda k: out[k]["value"])
    return {"value": out[better]["value"], "unit": "ref-bases/s", "ms": out[better]["ms"], "best": better,
            "workload": "synthetic 30x %s, %d reads, %.0f MB of records from host memory"
                        % ("WGS (3.1 Gb)" if genome else "chr20 (63 Mb)", n_reads, nbytes / 1e6),
            "includes": "host arrays -> pinned ring -> H2D + gd_compute on the records as they arrived + D2H of window "
                        "sums/minima and class runs; feed_ms: until the last block is committed (copies may still be in flight)",
            "host_to_device_GBps": out[better]["host_to_device_GBps"], "variants": out,
            "in_place_filler_threads": fillers}


def read_once(path, threads=16, block=8 << 20):
    """One pass over a file that was just written, before anything is timed on it: the first read after the write is when
    the kernel moves the file's pages to the active list of the page cache, under one lock taken by every reading thread
    -- measured on the GPU boxes as twice the system time and 3x the wall time of every later read of the same file
    (profiles/r11g_read_variance.jsonl: `pread` phase 0.54 s for 13 GB in the first run, 0.16-0.22 s in the next 25).
    Seconds it took."""
    from concurrent.futures import ThreadPoolExecutor
    t0 = time.perf_counter()
    size = os.path.getsize(path)
    fd = os.open(path, os.O_RDONLY)
    try:
        per = (size // threads + block) // block * block

        def part(i):
            buf = bytearray(block)
            off, end = i * per, min(size, (i + 1) * per)
            while off < end:
                n = os.preadv(fd, [buf], off)
                if n <= 0:
                    break
                off += n
        with ThreadPoolExecutor(threads) as ex:
            list(ex.map(part, range(threads)))
    finally:
        os.close(fd)
    return time.perf_counter() - t0


def expected_beds(which, W, chrom=None, live=False):
    """SHA-256 of the depth.bed / callable.bed that `goleft depth -w W` must write for the deterministic synthetic file
    `which`, FROM THE ORACLE (oracle/synthbam.py: numpy twin of the generator's record function -> oracle/depth_oracle.c ->
    the restated callback): the committed pair (tests/golden/synth_bam_expected.json, made by tools/expected_bed_sha.py on
    the CPU) or, live=True, computed here -- as the checker of a finished run, outside anything timed."""
    key = "%s:cov30:seed20:w%d%s" % (which, W, ":chrom=" + chrom if chrom else "")
    if live:
        from goleft_amd import synth
        from oracle import synthbam
        names = list(synth.HG19_NAMES)
        lengths = {"genome": list(synth.HG19_LENGTHS), "chr1-2": list(synth.HG19_LENGTHS[:2]),
                   "chr20-21": [synth.HG19_LENGTHS[names.index("chr20")], synth.HG19_LENGTHS[names.index("chr21")]]}[which]
        t0 = time.perf_counter()
        r = synthbam.expected_beds(lengths, W=W, chrom=chrom, threads=usable_cpus())
        return dict(r, source="oracle, computed in this run (%.1f s)" % (time.perf_counter() - t0), key=key)
    try:
        with open(os.path.join(ROOT, "tests", "golden", "synth_bam_expected.json")) as fh:
            r = json.load(fh).get(key)
    except (OSError, ValueError):
        r = None
    return dict(r, source="oracle, committed: tests/golden/synth_bam_expected.json", key=key) if r else None


def bam_file_scope(which, W, device_reps=3, host_decoder=True, pause_s=0.0, synth_env=None, paper=False, oracle_live=False, devices=None):
    """SURVEY.md section 8d scope (iii), the only scope the reference itself runs and times (`time goleft depth ...`,
    indexcov/paper/cmp.sh:6): a BAM FILE -> depth.bed + callable.bed through the CLI twin, process start to exit.
    A synthetic but realistic coordinate-sorted BAM (tools/synth_bam.cpp: 150 bp records WITH SEQ and QUAL, BGZF, .bai)
    of `which` -- "genome": all 24 hg19 contigs, 30x, ~46 GB; "chr1-2": two chromosomes, ~7 GB; "chr20-21": ~1.4 GB --
    is written to $TMPDIR, read once untimed (read_once above), then read by `goleft-depth depth -w W` with the device
    decoder (best of device_reps runs; the file is in the page cache, as after any write) and once with the host decoder;
    the BED files must be byte identical -- to each other AND to what the oracle makes of the same records
    (`oracle_identical`: expected_beds above).  synth_env: how the file is made (SYNTH_BAM_LEVEL: deflate level,
    SYNTH_BAM_AUX=1: Illumina-style names, mate fields and the tags of an aligned, duplicate-marked file).  paper: also the
    reference's own timed invocation, `goleft depth --chrom <first contig> -p 20 -o -w 16384` (indexcov/paper/cmp.sh:6), on
    the same file.  `pause_s`: seconds to wait before each run -- a process that starts while the
    driver is still clearing the device memory the previous one released waits for it in its first large hipMalloc
    (profiles/r11i_pause_test.jsonl, a 13 GB file: `gd_ingest_begin` 1.7-2.6 s and 2.5-3.4 s per run in 12 runs started
    back to back, 0.23-0.36 s and 1.06-1.26 s per run in 12 runs started 1, 2 or 4 s after the previous one's exit); one
    run of a real job has no such predecessor.  Reported next to -- never as -- `value`."""
    import hashlib
    import shutil
    import subprocess
    from goleft_amd import synth
    exe = os.path.join(ROOT, "goleft_amd", "goleft-depth")
    gen = os.path.join(ROOT, "goleft_amd", "synth-bam")
    names = list(synth.HG19_NAMES)
    lengths = {"genome": list(synth.HG19_LENGTHS), "chr1-2": list(synth.HG19_LENGTHS[:2]),
               "chr20-21": [synth.HG19_LENGTHS[names.index("chr20")], synth.HG19_LENGTHS[names.index("chr21")]]}[which]
    aux = bool(synth_env and synth_env.get("SYNTH_BAM_AUX") == "1")
    need = int(sum(lengths) * (20 if aux else 16))      # ~15 (19) B of BGZF per reference base at 30x, and the BED files
    tmp = os.environ.get("TMPDIR") or "/tmp"
    # a RAM-backed directory when it has the room: the GPU boxes' /tmp is an overlay that takes 0.3 GB/s (150 s for the
    # genome's file), and the file is read from the page cache either way
    if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > need * 1.3:
        tmp = "/dev/shm"
    free = shutil.disk_usage(tmp).free
    if free < need * 1.2:
        return {"error": "%s needs %.0f GB in %s, %.0f GB free" % (which, need / 1e9, tmp, free / 1e9)}
    d = tempfile.mkdtemp(prefix="gd_bamscope_", dir=tmp)

    def sha_pair(stem):
        h = []
        for kind in ("depth", "callable"):
            m = hashlib.sha256()
            with open("%s.%s.bed" % (stem, kind), "rb") as fh:
                for blk in iter(lambda: fh.read(1 << 24), b""):
                    m.update(blk)
            h.append(m.hexdigest())
        return h

    try:
        bam = os.path.join(d, "synth.bam")
        t0 = time.perf_counter()
        info = json.loads(subprocess.check_output([gen, bam, "chrS", ",".join(str(x) for x in lengths), "30", "20"],
                                                  timeout=900, env=dict(os.environ, **(synth_env or {}))).decode())
        t_write = time.perf_counter() - t0
        t_settle = read_once(bam)
        ref_bases = int(sum(lengths))
        out = {"what": "BAM file -> depth.bed + callable.bed, `goleft-depth depth -w %d -p 0 -r synth.fa --prefix OUT synth.bam`, "
                       "process start to exit (the scope the reference times: indexcov/paper/cmp.sh:6); file in the page cache "
                       "and read once before; %.1f s between runs" % (W, pause_s),
               "file": which, "contigs": len(lengths), "ref_bases": ref_bases, "reads": info["reads"],
               "bam_bytes": info["bam_bytes"], "deflate": info.get("deflate", "zlib level 1"),
               "records": info.get("records", "short names, no tags"), "inflated_bytes": info.get("inflated_bytes"),
               "synth_bam_s": t_write, "read_once_s": t_settle, "pause_before_each_run_s": pause_s,
               "written_to": tmp, "host_cores": os.cpu_count(), "usable_cpus": usable_cpus(), "unit": "ref-bases/s"}
        beds = {}
        # devices: GOLEFT_DEVICES for the CLI -- one engine context + worker thread per listed device, contigs by LPT, rows
        # merged in input order by the main thread (the in-process counterpart of the reference's -p pool,
        # depth/depth.go:392-421)
        dev_env = {"GOLEFT_DEVICES": devices} if devices else {}
        if devices:
            out["devices"] = devices
        runs = [("device", dev_env, device_reps)] + ([("host", {"GOLEFT_GPU_DECODE": "0"}, 1)] if host_decoder else [])

        def one_run(decoder, env, extra, stem):
            if pause_s:
                time.sleep(pause_s)
            t0 = time.perf_counter()
            p = subprocess.run([exe, "depth"] + extra + ["-r", os.path.join(d, "synth.fa"), "--prefix", stem, bam],
                               env=dict(os.environ, GOLEFT_DEPTH_TIMING="1", GOLEFT_INGEST_TIMING="1", **env),
                               stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, timeout=900)
            dt = time.perf_counter() - t0
            if p.returncode != 0:
                raise RuntimeError("goleft-depth (%s decoder) exited with %d: %s" % (decoder, p.returncode, p.stderr.decode()[-400:]))
            phases = {}
            for ln in p.stderr.decode().strip().splitlines():
                if ln.startswith("{"):
                    phases.update(json.loads(ln))
            if phases.get("decoder") != decoder:
                raise RuntimeError("asked for the %s decoder, %r ran" % (decoder, phases.get("decoder")))
            return dt, phases

        for decoder, env, reps in runs:
            walls, best, seen = [], None, []
            stem = os.path.join(d, "out_" + decoder)
            for _ in range(reps):
                try:
                    dt, phases = one_run(decoder, env, ["-w", str(W), "-p", "0"], stem)
                except RuntimeError as e:
                    return dict(out, error=str(e))
                walls.append(dt)
                seen.append({k: phases.get(k) for k in ("lib_begin_s", "lib_read_s", "lib_wait_link_s", "setup_s")})
                if best is None or dt < best[0]:
                    best = (dt, phases)
            beds[decoder] = sha_pair(stem)
            out[decoder + "_decoder"] = {"wall_s": best[0], "all_wall_s": walls, "all_runs": seen, "ref_bases_per_s": ref_bases / best[0],
                                         "bgzf_GBps": info["bam_bytes"] / best[0] / 1e9,
                                         "phases": {k: v for k, v in best[1].items() if isinstance(v, (int, float))}}
        out["value"] = out["device_decoder"]["ref_bases_per_s"]
        out["wall_s"] = out["device_decoder"]["wall_s"]
        out["bgzf_GBps"] = out["device_decoder"]["bgzf_GBps"]
        out["outputs_identical"] = (beds["device"] == beds["host"]) if "host" in beds else None
        out["bed_sha256"] = beds["device"]
        # ... and against the ORACLE's rows for the same records (not a second product path)
        exp = expected_beds(which, W, live=oracle_live) or (expected_beds(which, W, live=True) if sum(lengths) < 3e8 else None)
        out["oracle_identical"] = (exp["bed_sha256"] == beds["device"]) if exp else None
        out["oracle_bed_sha256"] = exp["bed_sha256"] if exp else None
        out["oracle_source"] = exp["source"] if exp else "no committed expectation for this file and too large to compute here"
        st = real_samtools()
        if st and sum(lengths) < 3e8:
            # a box WITH a samtools: the reference's own children on this very file, next to the numbers above
            try:
                out["reference_children"] = dict(reference_children(st[0], bam, [("chrS" if i == 0 else "chrS_%d" % (i + 1), L) for i, L in enumerate(lengths)],
                                                                    W, 1, usable_cpus()), samtools=st[1])
            except Exception as e:
                out["reference_children"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if paper:
            # the reference's own timed invocation (indexcov/paper/cmp.sh:6): one chromosome of a whole-genome BAM, found
            # through the .bai
            stem = os.path.join(d, "out_paper")
            W2 = 16384
            try:
                best = None
                for _ in range(2):
                    dt, phases = one_run("device", {}, ["--chrom", "chrS", "-p", "20", "-o", "-w", str(W2)], stem)
                    if best is None or dt < best[0]:
                        best = (dt, phases)
                got = sha_pair(stem + ".chrS")             # depth/depth.go:382-388: the prefix gains the chromosome
                exp2 = expected_beds(which, W2, chrom="chrS")
                out["paper_invocation"] = {
                    "what": "`goleft-depth depth --chrom chrS -p 20 -o -w 16384` on the same file: the invocation the reference "
                            "itself times (indexcov/paper/cmp.sh:6) -- a .bai seek + one chromosome (chr1-sized), process start to exit",
                    "wall_s": best[0], "ref_bases": int(lengths[0]), "ref_bases_per_s": lengths[0] / best[0],
                    "phases": {k: v for k, v in best[1].items() if isinstance(v, (int, float))},
                    "bed_sha256": got, "oracle_identical": (exp2["bed_sha256"] == got) if exp2 else None,
                    "oracle_source": exp2["source"] if exp2 else None}
                out["paper_invocation_s"] = best[0]
            except (RuntimeError, OSError) as e:
                out["paper_invocation"] = {"error": str(e)}
        return out
    finally:
        shutil.rmtree(d, ignore_errors=True)


def load_traffic(tag, kernel):
    """HBM bytes per launch of the dominant kernel(s) from the committed rocprofv3 PMC passes
    (profiles/*_<tag>_traffic.json: FETCH_SIZE / WRITE_SIZE collected in separate --pmc runs of this
    very command, tools/prof.sh + tools/traffic_from_pmc.py).  Only a file measured on the SAME
    kernel(s) this run timed is accepted: counter evidence of another kernel generation says nothing
    about this one."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_%s_traffic.json" % tag)), reverse=True):
        try:
            with open(path) as fh:
                d = json.load(fh)
        except (OSError, ValueError):
            continue
        if d.get("kernel") == kernel:               # the latest round's measurement of this kernel
            d["file"] = os.path.relpath(path, ROOT)
            return d
    return None


def run_case(args, scaling, world, rank, dev, local_rank, want_streams=False):
    """Set up one workload (weak: N-genome cohort, strong: one genome) on this
    rank, time `steps` steps, return a dict of measurements."""
    import torch
    import torch.distributed as dist
    from goleft_amd import shard, synth
    from goleft_amd.engine import DepthEngine, K_PREP, K_TILE, K_RUNS, K_EXPAND, K_SCAN, K_CKPT, K_NORM, TK_NAMES

    ont = args.workload.startswith("ont")
    cohort = args.workload == "cohort"
    if cohort:
        # BASELINE.json config 4: S samples x chr1, W = 250 windows -> depthwed matrix at -s 1000;
        # windows-only output (no per-base vector: 200 x 1 GB would not fit next to the records)
        S = args.samples
        names1 = ["s%03d.chr1" % k for k in range(S)]
        lengths1 = [synth.HG19_LENGTHS[0]] * S
        seeds1 = [5000 + k for k in range(S)]
        wname = "synthetic cohort: %d samples x chr1 (249 Mb), 150 bp reads, coverage 20-40x" % S
    elif args.workload in ("wgs", "ont"):
        names1, lengths1 = list(synth.HG19_NAMES), list(synth.HG19_LENGTHS)
        seeds1 = list(range(1, len(lengths1) + 1))
        wname = "synthetic 30x WGS, hg19 contig lengths (3.1 Gb), 150 bp reads"
    else:
        names1, lengths1, seeds1 = ["chr20"], [synth.CHR20_LEN], [20]
        wname = "synthetic 30x chr20 (63 Mb), 150 bp reads"
    if ont:
        wname = wname.replace("30x", "%gx ONT-like" % args.coverage).replace(
            "150 bp reads", "long reads (length-weighted median ~20 kb, ~1 CIGAR op per 13 aligned bases)")
    n_samples = world if scaling == "weak" else 1
    # the cohort is laid out as one reference of n_samples x contigs units
    names = ["s%d.%s" % (k, nm) for k in range(n_samples) for nm in names1] if n_samples > 1 else names1
    lengths = lengths1 * n_samples
    seeds = [1000 * k + sd for k in range(n_samples) for sd in seeds1]
    W, Q, mincov = args.window, 1, 4
    if cohort:
        W = 250 if args.window == 1000 else args.window     # goleft depth default window
    if scaling == "weak" and world > 1:
        # cohort: whole samples per rank (BASELINE.json config 4 "shards by sample"): every rank owns
        # the BED outputs of its genome, exactly as N independent `goleft depth` runs would -- no
        # exchange step exists, so none is invented
        per = len(lengths1)
        assignment = [list(range(k * per, (k + 1) * per)) for k in range(world)]
    else:
        assignment = shard.lpt_assign(lengths, world)
    mine = assignment[rank]
    exchange = world > 1 and scaling == "strong"            # one genome over N GPUs: rank 0 writes the BED

    # ---- synthetic record streams, generated on device, adopted zero-copy ----
    eng = DepthEngine(local_rank)
    eng.set_params(window_size=W, min_mapq=Q, min_cov=mincov)
    for kv in args.opt:
        k, v = kv.split("=")
        eng.set_option(int(k), int(v))
    if cohort:
        # depthwed only needs window sums: GD_OUT_SUMS_ONLY (read/window overlaps, no per-base scan);
        # --cohort-outputs windows keeps minima and class runs (the regular windows-only kernel)
        eng.set_outputs(perbase=False, sums_only=args.cohort_outputs == "sums")
    elif os.environ.get("GOLEFT_BENCH_OUTPUTS") == "windows":
        eng.set_outputs(perbase=False)     # experiment only (profiles/r01h): the tile kernel without its store stream
    eng.set_contigs(lengths)
    eng.select_contigs(mine)
    n_reads = n_ops = 0
    streams = {}
    for t in mine:
        if ont:
            n = synth.n_ont_reads_for(lengths[t], args.coverage)
            s = synth.ont_reads_torch(lengths[t], n, seeds[t], dev)
        elif cohort:
            cov = 20.0 + (seeds[t] * 7919 % 2001) / 100.0          # 20..40x, fixed per sample
            n = synth.n_reads_for(lengths[t], cov)
            s = synth.short_reads_torch(lengths[t], n, seeds[t], dev)
        else:
            n = synth.n_reads_for(lengths[t], args.coverage)
            s = synth.short_reads_torch(lengths[t], n, seeds[t], dev)
        eng.adopt_device(t, *s)
        streams[t] = s
        n_reads += n
        n_ops += int(s[4].shape[0])
    torch.cuda.synchronize()
    # derived structures (canonical records, deletion lists, tile indexes) are part of EVERY step where the path
    # needs them; the short-read tile path needs none
    derive = ont                                    # (the cohort's streaming sums read the records as they arrived)

    # ---- the COLD step: what ONE `goleft depth` run pays -- this context has never computed anything ----------
    # fresh context -> gd_adopt_device (above) -> ONE gd_compute with the default max_span_hint.  Device allocations
    # of the result arrays (gd_compute_timing's prepare share) are reported separately: `ms` is the rest.
    first = None
    if world == 1:
        eng.set_profiling(True)
        t1 = time.perf_counter()
        eng.compute()
        wall = time.perf_counter() - t1
        tm = eng.compute_timing()
        st1 = eng.stats()
        first = {"what": "fresh context, records adopted, ONE gd_compute (default max_span_hint): wall clock minus the "
                         "device allocations of the result arrays (prepare_alloc_ms), which only a context's first "
                         "compute of a job size pays",
                 "ms": (wall - tm["prepare_s"]) * 1e3, "wall_ms": wall * 1e3, "prepare_alloc_ms": tm["prepare_s"] * 1e3,
                 "enqueue_ms": tm["enqueue_s"] * 1e3, "wait_ms": tm["wait_s"] * 1e3,
                 "kernels_ms": {"prep": eng.kernel_ms(K_PREP), "tile": eng.kernel_ms(K_TILE), "runs": eng.kernel_ms(K_RUNS),
                                "long_read_structures": eng.kernel_ms(K_CKPT)},
                 "slow_tiles": int(st1.n_slow_tiles), "lookback": int(st1.lookback), "reruns": int(st1.reruns),
                 "max_span_seen": int(st1.max_span_seen), "kernel": TK_NAMES[int(st1.tile_kernel)]}
        eng.set_profiling(False)

    wed = {}
    gath = None
    if exchange:
        # set-up, outside the timed region: one compute to learn the boundary count, the ranks agree
        # on a fixed capacity, buffers are allocated once and the engine is told to fill the send
        # buffer itself (gd_set_export)
        eng.compute()
        # The collective a cgo host would call is the library's own (gd_comm_init / gd_gather_export: RCCL opened by the C
        # ABI, include/goleft_depth.h) -- the default since round 5; the 128-byte id travels through the process group
        # that exists anyway.  Every rank must take the same road: the ranks agree (MIN over "it came up here") and fall
        # back to torch.distributed's gather together when RCCL cannot be opened through the library anywhere
        # (GD_E_NODEVICE) or the communicator does not come up; GOLEFT_BENCH_NATIVE_GATHER=0 asks for the fallback.
        native = os.environ.get("GOLEFT_BENCH_NATIVE_GATHER", "1") != "0"
        native_note = "asked for" if not native else ""
        comm_lib = None
        if native:
            from goleft_amd.engine import comm_library, comm_unique_id
            fdev = dev if dist.get_backend() == "nccl" else "cpu"   # (dry runs over gloo: host tensors)
            ok, box = 1, [None]
            # EVERY rank asks, locally, whether the library can open RCCL here, and the ranks agree before any of them enters
            # the collective gd_comm_init: a rank that cannot would return at once and leave the others waiting in
            # ncclCommInitRank
            comm_lib = comm_library()
            if comm_lib is None:
                ok, native_note = 0, "gd_comm_library: RCCL cannot be opened through the library on rank %d" % rank
            flag = torch.tensor([ok], dtype=torch.int32, device=fdev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) and rank == 0:
                try:
                    box = [comm_unique_id()]
                except Exception as e:                  # GdError: RCCL not available through the library
                    ok, native_note = 0, "gd_comm_unique_id: %s" % e
            if int(flag.item()):
                flag = torch.tensor([ok], dtype=torch.int32, device=fdev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()):
                dist.broadcast_object_list(box, src=0)
                try:
                    eng.comm_init(rank, world, box[0])
                except Exception as e:
                    ok, native_note = 0, "gd_comm_init: %s" % e
                flag = torch.tensor([ok], dtype=torch.int32, device=fdev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if not int(flag.item()) and ok:
                    eng.comm_destroy()
            native = bool(int(flag.item()))
            if not native and not native_note:
                native_note = "another rank could not bring the library's communicator up"
        gath = shard.RootGather(assignment, lengths, W, rank, world, dev, bounds_cap=0, native=native)
        gath.reserve(eng.device_runs()[1])
        gath.attach(eng)
        gath.fallback_note = native_note
        gath.comm_lib = comm_lib
        if native:
            # one step through EACH collective before anything is timed: what rank 0 received through the library's
            # gather must be what torch.distributed's gather delivers for the same export blocks
            eng.compute(); gath.step_exported(); gath.drain()
            torch.cuda.synchronize()
            got = gath.recv.clone() if rank == 0 else None
            ref = shard.RootGather(assignment, lengths, W, rank, world, dev, bounds_cap=gath.cap_b, native=False)
            ref.attach(eng)
            eng.compute(); ref.step_exported(); ref.drain()
            torch.cuda.synchronize()
            same = 1
            if rank == 0:
                same = int(bool(torch.equal(got, ref.recv)))
            t = torch.tensor([same], dtype=torch.int32, device=dev if dist.get_backend() == "nccl" else "cpu")
            dist.broadcast(t, src=0)
            gath.native_verified = bool(int(t.item()))
            del ref
            gath.attach(eng)                            # (the export block points at this gather's buffers again)
            if not gath.native_verified:
                raise SystemExit("bench.py: gd_gather_export delivered other words than torch.distributed's gather")

    def step(inclusive=True):
        if derive and inclusive:
            eng.rebuild_derived()                   # deletion lists + tile indexes rebuilt from the records as they arrived, every step
        eng.compute()
        if cohort:
            # the sites x samples matrix of this rank's samples, left in HBM for its consumer
            # (the D2H of the whole matrix is timed separately, below -- never part of `value`)
            tids = np.asarray(mine, np.int32).reshape(-1, 1)
            wed["shape"] = [eng.depthwed_device(tids, args.wed_size)[1], len(mine)]
        if exchange:
            gath.step_exported()                    # ONE asynchronous collective on the buffer this compute filled;
                                                    # the next compute fills the other one (no allocation)

    def run_steps(n):
        if not exchange or n < 1:
            for _ in range(n):
                step()
            return
        # ONE genome over N GPUs: the steps are pipelined -- while the kernels of step k run, the host verifies
        # nothing (compute_finish(k - 1) already did), re-points the export block and issues the (asynchronous)
        # gather of step k - 1.  Every step is still a complete compute + gather; all of them have landed before
        # the clock stops (drain + synchronize below).
        if derive:
            eng.rebuild_derived()
        eng.compute_launch()
        for _ in range(n - 1):
            eng.compute_finish()
            gath.flip()
            if derive:
                eng.rebuild_derived()
            eng.compute_launch()
            gath.post()
        eng.compute_finish()
        gath.flip()
        gath.post()

    run_steps(args.warmup)
    if exchange:
        gath.drain()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(args.steps)
    if exchange:
        gath.drain()                                # the last steps' collectives have landed on rank 0
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    # kernel durations (HIP events on the engine's stream): the same steps again, profiled, after the timed region
    tile_ms, prep_ms, runs_ms, expand_ms, scan_ms, ckpt_ms, norm_ms = [], [], [], [], [], [], []
    for _ in range(args.steps):
        eng.set_profiling(True)                     # (also clears the accumulating NORM / CKPT timers)
        step()
        tile_ms.append(eng.kernel_ms(K_TILE))
        prep_ms.append(eng.kernel_ms(K_PREP))
        runs_ms.append(eng.kernel_ms(K_RUNS))
        expand_ms.append(eng.kernel_ms(K_EXPAND))
        scan_ms.append(eng.kernel_ms(K_SCAN))
        ckpt_ms.append(eng.kernel_ms(K_CKPT))
        norm_ms.append(eng.kernel_ms(K_NORM))
    eng.set_profiling(False)
    torch.cuda.synchronize()
    st_incl = eng.stats()
    incl_kernel, incl_canon, incl_slow = int(st_incl.tile_kernel), int(st_incl.n_canonical_ops), int(st_incl.n_slow_tiles)
    if int(st_incl.n_deletions) and not incl_canon:
        # deletion lists built straight from the records: a deletion is what an (M, N) pair of canonical ops was
        incl_canon = 2 * int(st_incl.n_deletions) + n_reads

    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    split = None
    if exchange:
        # where a step goes, measured AFTER the timed loop with a synchronisation between the two
        # halves (the timed loop itself has none): gd_compute wall, then the gather until it has landed
        tc, tg = [], []
        for _ in range(5):
            dist.barrier()
            torch.cuda.synchronize()
            a = time.perf_counter()
            eng.compute()
            b = time.perf_counter()
            gath.step_exported()
            gath.drain()                               # the (asynchronous) collective has landed
            torch.cuda.synchronize()
            c = time.perf_counter()
            tc.append(b - a)
            tg.append(c - b)
        loads = [sum(lengths[t] for t in assignment[r]) for r in range(world)]
        split = {"compute_ms_this_rank": float(np.median(tc)) * 1e3, "gather_ms_this_rank": float(np.median(tg)) * 1e3,
                 "shard_ref_bases": loads, "lpt_imbalance": max(loads) / (sum(loads) / world),
                 "lpt_speedup_ceiling": sum(loads) / max(loads),
                 "gather_bytes_per_rank": int(gath.total * 8), "bounds_capacity": int(gath.cap_b),
                 "collective": "gd_gather_export (RCCL from the C ABI)" if gath.native else "torch.distributed gather",
                 "collective_verified_against_torch_gather": getattr(gath, "native_verified", None),
                 "collective_fallback_reason": getattr(gath, "fallback_note", "") or None,
                 # which RCCL the library's collective used: the one the process had already (torch's own copy: ONE RCCL per process) or its own
                 "collective_library": ({"path": gath.comm_lib[0], "already_in_the_process": gath.comm_lib[1]} if getattr(gath, "comm_lib", None) else None),
                 "pipelined": "in the timed loop a step is finish(k-1); flip(); launch(k); post(): the asynchronous, "
                              "double-buffered gather of step k-1 and its host-side cost run under the kernels of step k; "
                              "compute_ms / gather_ms here are measured one after the other"}
        if rank == 0:
            g = gath.result()
            split["gather_overflow"] = bool(g["overflow"])
            split["boundaries_per_rank"] = g["true_counts"]
            # the gathered window sums are the whole genome's: one checksum the oracle side can repeat
            tot = 0
            for r_, p in enumerate(g["parts"]):
                tot += int(p[:gath.nwin[r_]].sum().item())
            split["gathered_sum_of_window_sums"] = tot

    # ---- the N-GPU job's shards, one after the other on this device (a measured stand-in for the scaling curve) ----
    emu = None
    if world == 1 and args.emulate_shards and not cohort:
        emu = {}
        for N in [int(x) for x in args.emulate_shards.split(",") if x.strip()]:
            if N < 2 or N > len(lengths):
                continue
            asg = shard.lpt_assign(lengths, N)
            per, gbytes = [], []
            for r_ in range(N):
                eng.select_contigs(asg[r_])
                if derive:
                    eng.drop_derived()              # a rank only ever holds its own contigs' structures (and their device blocks)
                nw_r = sum(shard.n_windows(lengths[t], W) for t in asg[r_])
                cap_b = 1 << 16
                buf = torch.zeros(1 + nw_r + (nw_r + 1) // 2 + cap_b, dtype=torch.int64, device=dev)
                eng.set_export(buf.data_ptr(), nw_r, cap_b)
                for _ in range(max(2, args.warmup)):
                    step()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    step()
                torch.cuda.synchronize()
                per.append((time.perf_counter() - t1) / args.steps * 1e3)
                gbytes.append(int(buf.numel() * 8))
                eng.set_export(0, 0, 0)
                del buf
            loads = [sum(lengths[t] for t in a) for a in asg]
            emu[str(N)] = {"per_shard_ms": per, "max_shard_ms": max(per), "shard_ref_bases": loads,
                           "projected_speedup": (dt / args.steps * 1e3) / max(per),
                           "lpt_speedup_ceiling": sum(loads) / max(loads), "gather_bytes_per_rank": gbytes}
        eng.select_contigs(mine)
        eng.compute()

    st = eng.stats()
    my_bases = sum(lengths[t] for t in mine)
    my_windows = sum(shard.n_windows(lengths[t], W) for t in mine)
    res = {
        "eng": eng, "streams": streams if want_streams else None, "names": names, "lengths": lengths,
        "mine": mine, "dt": dt, "total_bases": sum(lengths), "my_bases": my_bases,
        "my_windows": my_windows, "n_reads": n_reads, "n_ops": n_ops, "wname": wname,
        "n_samples": n_samples, "W": W, "Q": Q, "mincov": mincov,
        "tile_ms": float(np.mean(tile_ms)), "prep_ms": float(np.mean(prep_ms)),
        "runs_ms": float(np.mean(runs_ms)), "tile_positions": st.tile_positions, "lookback": st.lookback,
        "expand_ms": float(np.mean(expand_ms)), "scan_ms": float(np.mean(scan_ms)), "path": int(st.path),
        "ckpt_ms": float(np.mean(ckpt_ms)), "norm_ms": float(np.mean(norm_ms)), "derive": derive,
        "n_canonical_ops": incl_canon, "n_slow_tiles": incl_slow, "kernel": TK_NAMES[incl_kernel],
        "perbase": not cohort, "wed_shape": wed.get("shape"), "split": split, "emu": emu, "first": first,
    }
    if not want_streams:
        streams.clear()
    return res


def roofline_of(r, args, world):
    """The roofline of the dominant kernel of one run_case() result, this rank's launch.  SURVEY.md 8(d): 4 B/read (pos)
    + 4 B/read (the CSR offset, read on device) + 4 B per CIGAR op the kernel reads (the original ops on the short-read
    path; deletion lists = the canonical op pairs on the long-read path) + 4 B/base + 8 B/window.  `achieved` / `frac` are
    on that formula; the raw straight-line kernel really reads 3 B per read more (flag 2 + MAPQ 1): counting those too
    gives `frac_bytes_really_read`."""
    from goleft_amd import synth
    scatter = r["path"] == 2
    chunk = r["path"] == 3
    # SURVEY 8(d)'s op term is the ORIGINAL CIGAR stream: on the long-read path the step reads it once
    # (gd_dels_raw_kernel: deletion lists + tile index) and then the lists (gd_ltile2_kernel) -- two kernels share the work, so the
    # figure is taken over the INCLUSIVE step; each kernel's own figure is a sub-key
    ops_read = r["n_ops"] if chunk else (r["n_canonical_ops"] if r["n_canonical_ops"] else r["n_ops"])
    raw_records = r["kernel"].endswith("<raw>")
    alg_bytes = synth.algorithmic_bytes(r["n_reads"], ops_read, r["my_bases"] if r["perbase"] else 0,
                                        r["my_windows"], raw=False)         # windows-only: no 4 B/base write
    alg_bytes_read = synth.algorithmic_bytes(r["n_reads"], ops_read, r["my_bases"] if r["perbase"] else 0,
                                             r["my_windows"], raw=raw_records)
    # tile path: the tile kernel does all the arithmetic; chunk path: the inclusive step (the deletion lists and tile indexes
    # are rebuilt inside every step: `long_read_structures`, then the long-read tile kernel); scatter path: expand + scan share it
    step_s = r["dt"] / max(1, args.steps)
    avg_tile_s = max((r["expand_ms"] + r["scan_ms"]) * 1e-3 if scatter else step_s if chunk else r["tile_ms"] * 1e-3, 1e-9)   # (a rank without contigs: nothing ran)
    achieved = alg_bytes / avg_tile_s / 1e9
    traffic = None
    tr = None
    kname = r["kernel"]                             # gd_stats.tile_kernel: what did the per-base arithmetic
    if world == 1 and args.workload == "wgs" and args.coverage == 30.0 and r["path"] == 1:
        tr = load_traffic("wgs", kname)
    elif world == 1 and args.workload == "ont" and args.coverage == 20.0 and chunk:
        # the step's two kernels: the counters of both, summed (each file is accepted only for the kernel it names)
        tr = load_traffic("ont", kname)
        tr2 = load_traffic("ont_dels_raw", "gd_dels_raw_kernel")
        if tr and tr2:
            tr = dict(tr, hbm_bytes_per_launch=tr["hbm_bytes_per_launch"] + tr2["hbm_bytes_per_launch"],
                      file=tr["file"] + " + " + tr2["file"], per_kernel={kname: tr["hbm_bytes_per_launch"], "gd_dels_raw_kernel": tr2["hbm_bytes_per_launch"]})
        else:
            tr = None
    elif world == 1 and args.workload == "cohort" and args.samples == 200 and kname.startswith("gd_sums_stream_kernel"):
        tr = load_traffic("cohort", kname)
    elif world == 1 and args.workload == "chr20" and args.coverage == 30.0 and r["path"] == 1:
        tr = load_traffic("chr20", kname)
    if tr:
        traffic = tr.get("hbm_bytes_per_launch")   # measured on this exact launch shape
    roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                # MI355X_MICROARCH.md (HBM): 8 TB/s is the spec, ~6.3 TB/s what a streaming kernel can reach
                "achievable": HBM_ACHIEVABLE_GBPS, "frac_of_achievable": achieved / HBM_ACHIEVABLE_GBPS,
                "formula": "SURVEY.md 8(d): 4 B/read pos + 4 B/read CSR offset + 4 B/op + 4 B/ref-base + 8 B/window",
                "frac_survey_8d": achieved / HBM_PEAK_GBPS,
                "frac_bytes_really_read": alg_bytes_read / avg_tile_s / 1e9 / HBM_PEAK_GBPS,
                "bytes_really_read_per_launch": alg_bytes_read,
                "kernel": "gd_dels_raw_kernel + gd_ltile2_kernel (the inclusive step)" if chunk else kname,
                "avg_kernel_ms": avg_tile_s * 1e3,
                "algorithmic_bytes_per_launch": alg_bytes,
                "bytes_per_ref_base": alg_bytes / max(1, r["my_bases"]),
                "cigar_ops_counted": ops_read}
    if chunk:
        # the two kernels on their own bytes: the list pass reads 8 B/read + the original ops and writes the lists; the tile
        # kernel reads the lists (the canonical op pairs) and writes the per-base vector and the windows
        dels_s, lt_s = max(r["ckpt_ms"] * 1e-3, 1e-9), max(r["tile_ms"] * 1e-3, 1e-9)
        dels_bytes = 8 * r["n_reads"] + 4 * r["n_ops"]
        lt_bytes = synth.algorithmic_bytes(r["n_reads"], r["n_canonical_ops"] or r["n_ops"], r["my_bases"] if r["perbase"] else 0, r["my_windows"], raw=False)
        roofline["kernels"] = {
            "gd_dels_raw_kernel": {"ms": r["ckpt_ms"], "bytes": dels_bytes, "what": "8 B/read + 4 B per ORIGINAL op read (the lists it writes are not counted)",
                                   "frac": dels_bytes / dels_s / 1e9 / HBM_PEAK_GBPS},
            "gd_ltile2_kernel": {"ms": r["tile_ms"], "bytes": lt_bytes, "what": "8 B/read + 4 B per deletion-list entry + 4 B/ref-base + 8 B/window",
                                 "cigar_ops_counted": r["n_canonical_ops"] or r["n_ops"], "frac": lt_bytes / lt_s / 1e9 / HBM_PEAK_GBPS}}
    if traffic:
        roofline["traffic_frac_of_peak"] = traffic / avg_tile_s / 1e9 / HBM_PEAK_GBPS
        roofline["traffic_source"] = "%s: %s" % (tr.get("file"), tr.get("source"))
    kernels_ms = ({"prep": r["prep_ms"], "expand": r["expand_ms"], "scan": r["scan_ms"], "runs": r["runs_ms"]}
                  if scatter else
                  {"prep": r["prep_ms"], "ltile": r["tile_ms"], "runs": r["runs_ms"], "long_read_structures": r["ckpt_ms"]}
                  if chunk else {"prep": r["prep_ms"], "tile": r["tile_ms"], "runs": r["runs_ms"]})
    return {"roofline": roofline, "kernels_ms": kernels_ms, "raw_records": raw_records, "scatter": scatter, "chunk": chunk,
            "avg_tile_s": avg_tile_s, "traffic": traffic, "tr": tr, "kernel": kname, "ops_read": ops_read,
            "alg_bytes": alg_bytes, "alg_bytes_read": alg_bytes_read, "achieved": achieved}


def other_workloads(args, dev, local_rank, which, steps=5, warmup=2):
    """BASELINE.json's configs 2, 4 and 5 in the line the driver records: after the headline, a short
    run of each -- the same run_case() as `--workload chr20|ont|cohort`, fewer steps -- with its inclusive step, the cold
    step (`first_compute`), the roofline of its dominant kernel on its OWN SURVEY 8(d) bytes and the kernel split.
    One at a time, each with the device to itself (the cohort's records alone are 154 GB)."""
    import copy
    import torch
    out = {}
    for w in which:
        t0 = time.perf_counter()
        a = copy.copy(args)
        a.workload, a.steps, a.warmup, a.emulate_shards, a.verify = w, steps, warmup, "", False
        a.coverage = 20.0 if w.startswith("ont") else 30.0
        a.window = 1000
        try:
            r = run_case(a, "strong", 1, 0, dev, local_rank)
            rf = roofline_of(r, a, 1)
            ms = r["dt"] / a.steps * 1e3
            sub = {"metric": ("ref bases/sec per-base depth, 20x ONT-like synthetic" if w.startswith("ont")
                              else "ref bases/sec depth -> depthwed matrix, cohort x chr1" if w == "cohort"
                              else "ref bases/sec per-base depth, 30x chr20 synthetic"),
                   "baseline_config": {"chr20": 2, "cohort": 4, "ont": 5}.get(w),
                   "workload": r["wname"], "value": r["total_bases"] * a.steps / r["dt"], "unit": "ref-bases/s",
                   "ms_per_step": ms, "steps": a.steps, "warmup": a.warmup,
                   "total_ref_bases": r["total_bases"], "reads": r["n_reads"], "cigar_ops": r["n_ops"],
                   "window": r["W"], "roofline": rf["roofline"], "kernels_ms": rf["kernels_ms"],
                   "device_path": "scatter" if rf["scatter"] else "chunk" if rf["chunk"] else "tile",
                   "step": ("gd_rebuild_derived (deletion lists, read records, tile indexes from the records as they arrived) "
                            "+ gd_compute" if r["derive"] else "gd_compute on the records as they arrived") +
                           (" + gd_depthwed_device" if w == "cohort" else "")}
            if r.get("first"):
                sub["first_compute"] = dict(r["first"], warm_ms_per_step=ms, ratio_to_warm=r["first"]["ms"] / ms)
                # what ONE `goleft depth` run pays for its input: the cold step, not the warm one
                sub["value_first_compute"] = r["total_bases"] / (r["first"]["ms"] * 1e-3)
            if r.get("wed_shape"):
                sub["depthwed_matrix"] = r["wed_shape"]
            r["eng"].close()
            del r
        except Exception as e:                           # never lose the headline line to a side measurement
            sub = {"error": "%s: %s" % (type(e).__name__, e)}
        torch.cuda.empty_cache()
        sub["seconds_in_bench"] = time.perf_counter() - t0
        out[w] = sub
    return out


def flush_c_stdio():
    """fflush(NULL): what C libraries of this process wrote with stdio -- RCCL prints a version banner (`ROCm version : ...`,
    `Librccl path : ...`) through it when it is loaded -- leaves the process buffer now instead of at exit, where it would
    land BEHIND the JSON line on a stdout that is a pipe (the driver reads the line from there)."""
    try:
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment: THIS process is the launcher.
    It starts N copies of itself, one rank per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 /
    MASTER_PORT set, everything else inherited -- the environment `torch.distributed.run` would give them), waits
    for all of them and exits with the first non-zero status.  It refuses to start when the node has fewer than N
    devices: N ranks on fewer GPUs is not an N-GPU measurement (the reference's counterpart is the `-p` worker
    pool, depth/depth.go:392-394).  GOLEFT_BENCH_SINGLE_DEVICE=1 (dry run, every rank on device 0) and
    GOLEFT_BENCH_STUB=1 (CPU test of this launcher) lift that check and say so in the line."""
    import socket
    import subprocess
    n = args.gpus
    stub = os.environ.get("GOLEFT_BENCH_STUB") == "1"
    if not stub:
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if os.environ.get("GOLEFT_BENCH_SINGLE_DEVICE") == "1":
            if have < 1:
                raise SystemExit("bench.py --gpus %d (single-device dry run): no GPU visible" % n)
        elif have < n:
            raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible on this node -- refusing to run %d ranks on "
                             "fewer devices (that would not be an %d-GPU measurement)" % (n, have, n, n))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GOLEFT_BENCH_LAUNCHER="self")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    alive = set(range(n))
    while alive:
        for r in sorted(alive):
            c = procs[r].poll()
            if c is None:
                continue
            alive.discard(r)
            if c != 0 and rc == 0:
                rc = c if c > 0 else 1
                sys.stderr.write("bench.py: rank %d exited with status %d; stopping the other ranks\n" % (r, c))
                for q in alive:
                    procs[q].terminate()                # exactly the processes started above
        if alive:
            time.sleep(0.05)
    raise SystemExit(rc)


def stub_rank(args, world, rank):
    """GOLEFT_BENCH_STUB=1: the launcher and the rendezvous without an engine (tests/test_bench_launcher.py).
    Every rank joins a gloo group, runs `steps` trivial steps between the same barriers as the real loop and
    rank 0 prints a line that cannot be mistaken for a measurement (value null, data "stub")."""
    import torch
    import torch.distributed as dist
    if os.environ.get("GOLEFT_BENCH_STUB_FAIL_RANK") == str(rank):
        raise SystemExit(7)                             # (the test of the launcher's failure handling)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    seen = who_took_part(world, rank, "cpu:%d" % os.getpid())
    acc = np.zeros(16, np.int64)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        acc += k
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(dict({"metric": "stub (launcher test, nothing measured)", "value": None, "unit": "ref-bases/s",
                               "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                               "ms_per_step": dt / max(1, args.steps) * 1e3, "data": "stub"}, **seen)))


def who_took_part(world, rank, my_device):
    """The ranks and devices that really joined the process group, collected through it: `n_gpus` in the line is
    only believable when `ranks_seen` == --gpus and `devices_seen` names that many DIFFERENT devices."""
    if world == 1:
        return {"ranks_seen": 1, "devices_seen": [my_device], "distinct_devices": 1,
                "launcher": os.environ.get("GOLEFT_BENCH_LAUNCHER", "none")}
    import torch.distributed as dist
    got = [None] * world
    dist.all_gather_object(got, (rank, my_device))
    assert dist.get_world_size() == world
    ranks = sorted(g[0] for g in got)
    devs = [g[1] for g in sorted(got)]
    return {"ranks_seen": len(set(ranks)), "devices_seen": devs, "distinct_devices": len(set(devs)),
            "launcher": os.environ.get("GOLEFT_BENCH_LAUNCHER", "torch.distributed.run or another external launcher")}


def device_identity(torch, index):
    p = torch.cuda.get_device_properties(index)
    u = getattr(p, "uuid", None)
    bus = "%04x:%02x:%02x" % (getattr(p, "pci_domain_id", 0), getattr(p, "pci_bus_id", 0), getattr(p, "pci_device_id", 0))
    return "%s@%s#%d" % (u if u is not None else p.name, bus, index)


def main():
    args = parse()
    if args.coverage is None:
        args.coverage = 20.0 if args.workload.startswith("ont") else 30.0
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        launch_ranks(args)                              # never returns
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        # an external launcher started a different number of ranks than --gpus asks for: the line's n_gpus would
        # contradict the command line either way
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if os.environ.get("GOLEFT_BENCH_STUB") == "1":
        return stub_rank(args, world, rank)
    import torch
    import torch.distributed as dist
    from goleft_amd import synth

    # dry-run hooks for the 1-GPU dev box (the multi-process control flow with every rank on
    # device 0 over gloo); the driver's real runs use one GPU per rank over RCCL
    backend = os.environ.get("GOLEFT_BENCH_BACKEND", "nccl")
    single = os.environ.get("GOLEFT_BENCH_SINGLE_DEVICE") == "1"
    if single:
        local_rank = 0
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the engine has no CPU fallback)")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d wants device %d but only %d GPU(s) are visible"
                         % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    took_part = who_took_part(world, rank, device_identity(torch, local_rank))
    if single and world > 1:
        took_part["launcher"] += " (GOLEFT_BENCH_SINGLE_DEVICE=1: every rank on device 0 -- a dry run of the control flow, not a measurement)"
    if world > 1 and not single and took_part["distinct_devices"] != world:
        raise SystemExit("bench.py --gpus %d: the %d ranks sit on %d distinct device(s) %s -- not an %d-GPU run"
                         % (args.gpus, world, took_part["distinct_devices"], took_part["devices_seen"], world))

    r = run_case(args, args.scaling, world, rank, dev, local_rank, want_streams=True)
    eng, streams, names, lengths, mine = r["eng"], r["streams"], r["names"], r["lengths"], r["mine"]
    dt, W, Q, mincov = r["dt"], r["W"], r["Q"], r["mincov"]
    value = r["total_bases"] * args.steps / dt
    rf = roofline_of(r, args, world)
    raw_records, scatter, chunk = rf["raw_records"], rf["scatter"], rf["chunk"]
    avg_tile_s, traffic, tr, kname = rf["avg_tile_s"], rf["traffic"], rf["tr"], rf["kernel"]
    ops_read, alg_bytes, alg_bytes_read, achieved = rf["ops_read"], rf["alg_bytes"], rf["alg_bytes_read"], rf["achieved"]

    # PCIe-inclusive rate (results to host) -- reported, never `value`
    t1 = time.perf_counter()
    sums_only = args.workload == "cohort" and args.cohort_outputs == "sums"
    for t in mine[:48]:
        eng.window_sums(t) if sums_only else eng.windows(t)
    d2h = (time.perf_counter() - t1) * (len(mine) / max(1, min(len(mine), 48)))
    d2h_matrix = None
    if args.workload == "cohort":
        t1 = time.perf_counter()
        eng.depthwed(np.asarray(mine, np.int32).reshape(-1, 1), args.wed_size)
        d2h_matrix = time.perf_counter() - t1

    out = {
        "metric": ("ref bases/sec per-base depth, 20x ONT-like synthetic" if args.workload.startswith("ont")
                   else "ref bases/sec depth -> depthwed matrix, cohort x chr1" if args.workload == "cohort"
                   else "ref bases/sec per-base depth, 30x WGS synthetic"),
        "value": value,
        "unit": "ref-bases/s",
        "n_gpus": world,
        "ranks_seen": took_part["ranks_seen"],
        "devices_seen": took_part["devices_seen"],
        "distinct_devices": took_part["distinct_devices"],
        "launcher": took_part["launcher"],
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "int32",
        "data": "synthetic",
        "config": {"workload": r["wname"] + (" x %d samples (cohort, one genome per GPU)" % r["n_samples"]
                                             if r["n_samples"] > 1 else
                                             ", ONE genome sharded by chromosome (LPT) over %d GPUs, gather to rank 0 "
                                             "inside the timed region" % world if world > 1 else ""),
                   "coverage": args.coverage, "window": W,
                   "min_mapq": Q, "min_cov": mincov, "total_ref_bases": r["total_bases"],
                   "reads_rank0": r["n_reads"], "cigar_ops_rank0": r["n_ops"],
                   "canonical_cigar_ops_rank0": r["n_canonical_ops"], "tiles_on_the_generic_kernel": r["n_slow_tiles"],
                   "step": ("gd_rebuild_derived [deletion lists, read records and tile indexes rebuilt straight from the "
                            "records as they arrived] + gd_compute" if r["derive"] else
                            "gd_compute on the records as they arrived (nothing derived exists)" if raw_records else
                            "gd_compute") + (" + gd_depthwed_device" if args.workload == "cohort" else ""),
                   "in_step_normalise_kernels_ms_rank0": r["norm_ms"], "in_step_checkpoint_kernels_ms_rank0": r["ckpt_ms"],
                   "sharding": ("single GPU" if world == 1 else
                                "by sample (one genome per GPU, no exchange)" if args.scaling == "weak" else
                                "by chromosome, LPT, RCCL gather of window sums/minima + class runs to rank 0"),
                   "outputs": ("(sums-only) int64 window sums" if (args.workload == "cohort" and args.cohort_outputs == "sums") else
                               ("int32 per-base depth + " if r["perbase"] else "(windows-only) ") +
                               "int64/int32 window sum/min + class runs") +
                              (" + depthwed matrix %s" % r["wed_shape"] if r["wed_shape"] else ""),
                   "tile_positions": r["tile_positions"], "lookback": r["lookback"],
                   "device_path": "scatter" if scatter else "chunk" if chunk else "tile"},
        "roofline": rf["roofline"],
        "kernels_ms": rf["kernels_ms"],
        "with_d2h_windows_ref_bases_per_s": r["my_bases"] / (dt / args.steps + d2h) if world == 1 else None,
    }
    if r.get("first"):
        out["first_compute"] = dict(r["first"], warm_ms_per_step=dt / args.steps * 1e3,
                                    ratio_to_warm=r["first"]["ms"] / (dt / args.steps * 1e3))
    if r.get("emu"):
        out["emulated_sharding"] = {
            "what": "every LPT shard of the N-GPU job (BASELINE.json config 3: one genome, contigs by LPT) computed on THIS "
                    "one device, one shard after the other: the same step as the headline incl. the export block every rank "
                    "hands to the gather; projected_speedup = step(N=1) / slowest shard.  The gather itself (one "
                    "asynchronous collective of gather_bytes_per_rank per rank to rank 0, issued under the next step's "
                    "kernels) is not in it: no second GPU here",
            "by_n_gpus": r["emu"]}
    if world == 1 and args.workload != "cohort":
        from goleft_amd import shard as _sh
        # checksum of checksums: equals "split.gathered_sum_of_window_sums" of an N > 1 run of the same workload
        out["sum_of_window_sums"] = int(_sh.local_results(eng, dev)[0].sum().item())
    if d2h_matrix is not None:
        out["with_d2h_matrix_ref_bases_per_s"] = r["my_bases"] / (dt / args.steps + d2h_matrix)
    if args.verify and rank == 0 and r["perbase"]:
        # EVERY contig of this rank against the C oracle, bit for bit: per-base vector, window sums / minima and
        # class runs (from the oracle's vector), 10 Mb tiles on all host cores
        from oracle import pyoracle as po
        cores = usable_cpus()
        t1 = time.perf_counter()
        ok, bad = True, []
        for t in mine:
            a = [x.cpu().numpy() for x in streams[t]]
            rd = po.Reads(a[0], a[1].view(np.uint16), a[2], a[3].view(np.uint32), a[4].view(np.uint32))
            res = po.tiled_contig_check(rd, Q, lengths[t], W, mincov, 0, po.step_for(W), cores, got=eng.perbase(t))
            sums, mins = eng.windows(t)
            runs = eng.callable_runs(t)
            good = (res["equal"] and np.array_equal(sums, res["sums"]) and np.array_equal(mins, res["mins"]) and
                    np.array_equal(runs[:, 0], res["run_starts"]) and np.array_equal(runs[:, 2], res["run_cls"]))
            if not good:
                ok = False
                bad.append(names[t])
            del a, rd, res
        out["verified_contigs"] = len(mine)
        out["verified_bit_exact"] = ok
        out["verified_seconds"] = time.perf_counter() - t1
        if bad:
            out["verified_mismatch"] = bad

    if rank == 0 and world == 1 and not args.no_host_stream and args.workload in ("wgs", "chr20"):
        out["host_stream_scope"] = host_stream_scope(local_rank, W, Q, mincov, opts=args.opt)
        if args.workload == "wgs":
            # the genome-sized run needs the HBM the resident streams hold: measured after they are released (below)
            out["host_stream_scope_wgs"] = "pending"

    if rank == 0 and world == 1 and not args.no_cpu_baseline:      # a reported baseline, N = 1 only
        from oracle import pyoracle as po
        cores = usable_cpus()                     # threads actually used: what the container's CPU quota lets run at once
        sample = []
        for t in mine[:args.cpu_sample_contigs]:
            a = [x.cpu().numpy() for x in streams[t]]
            rd = po.Reads(a[0], a[1].view(np.uint16), a[2], a[3].view(np.uint32), a[4].view(np.uint32))
            sample.append((names[t], lengths[t], rd))
        best = None
        for _ in range(2):                       # best of two passes (thread start-up noise)
            v, b, sec = cpu_baseline(sample, W, mincov, cores)
            if best is None or v > best[0]:
                best = (v, b, sec)
        v, b, sec = best
        out["cpu_baseline"] = {"value": v, "unit": "ref-bases/s", "cores": cores, "host_cpus": os.cpu_count(), "kind": "port",
                               "sample": "%s (%d ref bases) of the same stream, oracle/depth_oracle.c "
                                         "perbase_diff + callback, %d threads over 10 Mb tiles, %.1f s "
                                         "wall (%.0f core-seconds)"
                                         % ("+".join(s[0] for s in sample), b, cores, sec, sec * cores)}
        # the same port on ONE thread, on the last contig of the sample
        v1, b1, sec1 = cpu_baseline(sample[-1:], W, mincov, 1)
        out["cpu_baseline"]["single_thread"] = {
            "value": v1, "unit": "ref-bases/s", "cores": 1,
            "sample": "%s (%d ref bases), one thread, %.1f s" % (sample[-1][0], b1, sec1)}
    r_split = r["split"]
    eng.close()
    streams.clear()
    del r
    if out.get("host_stream_scope_wgs") == "pending":
        torch.cuda.empty_cache()
        try:
            out["host_stream_scope_wgs"] = host_stream_scope(local_rank, W, Q, mincov, genome=True, reps=1, opts=args.opt)
        except Exception as e:                       # never lose the headline line to a side measurement
            out["host_stream_scope_wgs"] = {"error": "%s: %s" % (type(e).__name__, e)}

    if rank == 0 and world == 1 and args.workload == "wgs" and args.other_workloads.strip():
        torch.cuda.empty_cache()
        which_w = [w for w in (x.strip() for x in args.other_workloads.split(",")) if w in ("chr20", "ont", "ont-chr20", "cohort")]
        out["other_workloads"] = other_workloads(args, dev, local_rank, which_w)

    if rank == 0 and world == 1 and args.workload == "wgs" and args.bam_scope != "off":
        which = args.bam_scope
        if which == "auto":
            which = "genome" if (os.cpu_count() or 1) >= 64 else "chr20-21"
        torch.cuda.empty_cache()
        try:
            # the genome-sized file with the device decoder (the number); the host decoder -- 39 s per genome on this
            # class of box, profiles/r04f_bench_wgs_n1.json -- runs on a small file of the same make, where the two
            # decoders' BED files are compared byte for byte
            # (the pause: 1 s is enough after a run over a 13 GB file, profiles/r11i_pause_test.jsonl; the genome's run
            # releases four times the device memory)
            res = bam_file_scope(which, W, device_reps=3, host_decoder=which != "genome", pause_s=8.0 if which == "genome" else 2.0,
                                 paper=which == "genome")
            if "error" in res and which == "genome" and args.bam_scope == "auto":
                res = dict(bam_file_scope("chr1-2", W), fell_back_from=res["error"])
            if which == "genome" and "error" not in res:
                # a small file of the same make, written two ways -- as the genome's (libdeflate level 1, the cheapest records a
                # BAM can hold) and as an aligner + duplicate marker + htslib leave one (level 6; names, mate fields, tags) --
                # read by BOTH decoders: BED files byte identical to each other and to the oracle's rows
                res["variants"] = {}
                for vname, venv in (("libdeflate1_short_records", {}),
                                    ("libdeflate6_aux_tags", {"SYNTH_BAM_LEVEL": "6", "SYNTH_BAM_AUX": "1"})):
                    chk = bam_file_scope("chr20-21", W, device_reps=2, host_decoder=True, pause_s=2.0, synth_env=venv,
                                         oracle_live=vname == "libdeflate1_short_records")
                    v = {k: chk.get(k) for k in ("file", "ref_bases", "bam_bytes", "inflated_bytes", "deflate", "records", "outputs_identical",
                                                 "oracle_identical", "oracle_source", "bed_sha256", "error") if k in chk}
                    v["device_wall_s"] = (chk.get("device_decoder") or {}).get("wall_s")
                    v["host_wall_s"] = (chk.get("host_decoder") or {}).get("wall_s")
                    v["device_ref_bases_per_s"] = (chk.get("device_decoder") or {}).get("ref_bases_per_s")
                    v["device_phases"] = (chk.get("device_decoder") or {}).get("phases")
                    if "reference_children" in chk:
                        v["reference_children"] = chk["reference_children"]
                    res["variants"][vname] = v
                # ... and the GENOME written the second way: what a 30x file costs as an aligner leaves it
                try:
                    big = bam_file_scope("genome", W, device_reps=2, host_decoder=False, pause_s=8.0,
                                         synth_env={"SYNTH_BAM_LEVEL": "6", "SYNTH_BAM_AUX": "1"})
                    dd = big.get("device_decoder") or {}
                    res["genome_libdeflate6_aux_tags"] = {k: big.get(k) for k in ("file", "ref_bases", "bam_bytes", "inflated_bytes", "deflate", "records", "oracle_identical",
                                                                                     "oracle_source", "bed_sha256", "synth_bam_s", "error") if k in big}
                    res["genome_libdeflate6_aux_tags"].update({"wall_s": dd.get("wall_s"), "all_wall_s": dd.get("all_wall_s"), "ref_bases_per_s": dd.get("ref_bases_per_s"),
                                                               "phases": dd.get("phases")})
                except Exception as e:                   # never lose the headline line to a side measurement
                    res["genome_libdeflate6_aux_tags"] = {"error": "%s: %s" % (type(e).__name__, e)}
                first = res["variants"]["libdeflate1_short_records"]
                res["decoders_identical_on"] = {k: first.get(k) for k in ("file", "ref_bases", "bam_bytes", "outputs_identical",
                                                                           "oracle_identical", "device_wall_s", "host_wall_s", "error") if k in first}
                res["outputs_identical"] = all(v.get("outputs_identical") is True for v in res["variants"].values())
                res["oracle_identical"] = (res.get("oracle_identical") is True and
                                           all(v.get("oracle_identical") is True for v in res["variants"].values()))
            out["bam_file_scope"] = res
            # a box with a samtools: the reference's children, timed on the small file, next to the port
            rc = ((res.get("variants") or {}).get("libdeflate1_short_records") or {}).get("reference_children") or res.get("reference_children")
            out.setdefault("cpu_baseline", {})["reference_pipeline"] = rc if rc else {
                "available": False, "why": "no samtools on PATH (or $SAMTOOLS) on this box; bench.py times the reference's children by itself where there is one"}
        except Exception as e:                       # never lose the headline line to a side measurement
            out["bam_file_scope"] = {"error": "%s: %s" % (type(e).__name__, e)}

    if r_split is not None:
        out["split"] = r_split
    # N > 1: the headline is ONE genome over N GPUs (BASELINE.json config 3); the cohort case (N genomes,
    # one per GPU, no exchange) is reported next to it
    if world > 1 and args.scaling == "strong" and args.workload in ("wgs", "ont") and \
            not os.environ.get("GOLEFT_BENCH_SKIP_COHORT"):     # (dry runs on the 1-GPU box skip the side case)
        try:
            torch.cuda.empty_cache()
            r2 = run_case(args, "weak", world, rank, dev, local_rank)
            out["cohort_weak_scaling"] = {"value": r2["total_bases"] * args.steps / r2["dt"], "unit": "ref-bases/s",
                                          "ms_per_step": r2["dt"] / args.steps * 1e3,
                                          "total_ref_bases": r2["total_bases"],
                                          "workload": r2["wname"] + " x %d samples, one genome per GPU, no exchange" % world,
                                          "kernels_ms_rank0": {"prep": r2["prep_ms"], "tile": r2["tile_ms"],
                                                               "runs": r2["runs_ms"]}}
            r2["eng"].close()
        except Exception as e:                       # never lose the headline line to the secondary case
            out["cohort_weak_scaling"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if world > 1:
        torch.cuda.empty_cache()
        dist.barrier()                                  # every rank has closed its engine: the devices are free
        file_ngpu = args.workload == "wgs" and args.scaling == "strong" and not single and \
            os.environ.get("GOLEFT_BENCH_SKIP_FILE_NGPU") != "1"
        store = None
        if file_ngpu:
            try:
                store = dist.distributed_c10d._get_default_store()   # the ranks wait on the HOST (a barrier would spin a kernel on every device)
            except Exception:
                store = None
        if rank == 0 and file_ngpu:
            # the file scope over the node's N devices: ONE `goleft-depth` process, GOLEFT_DEVICES=0..N-1 (what a user of
            # an N-GPU node runs); BED files checked against the oracle's.  The other ranks wait at the barrier below.
            try:
                torch.cuda.empty_cache()
                which = "genome" if (os.cpu_count() or 1) >= 64 else "chr20-21"
                res = bam_file_scope(which, W, device_reps=2, host_decoder=False, pause_s=8.0 if which == "genome" else 2.0,
                                     devices=",".join(str(i) for i in range(world)))
                out["bam_file_scope_ngpu"] = {k: res.get(k) for k in ("what", "file", "devices", "ref_bases", "bam_bytes", "deflate", "records",
                                                                       "wall_s", "value", "unit", "bgzf_GBps", "bed_sha256", "oracle_identical",
                                                                       "oracle_source", "synth_bam_s", "error") if k in res}
                out["bam_file_scope_ngpu"]["all_wall_s"] = (res.get("device_decoder") or {}).get("all_wall_s")
                out["bam_file_scope_ngpu"]["phases"] = (res.get("device_decoder") or {}).get("phases")
            except Exception as e:                       # never lose the headline line to a side measurement
                out["bam_file_scope_ngpu"] = {"error": "%s: %s" % (type(e).__name__, e)}
            if store is not None:
                store.set("goleft_file_ngpu_done", "1")
        elif file_ngpu and store is not None:
            import datetime
            try:
                store.wait(["goleft_file_ngpu_done"], datetime.timedelta(minutes=30))
            except Exception:
                pass
        flush_c_stdio()                                 # (RCCL's version banner sits in the C library's buffer: out with it NOW,
        dist.barrier()                                  #  on every rank, before rank 0 prints the line -- which must come last)
        dist.destroy_process_group()
    flush_c_stdio()
    if rank == 0:
        print(json.dumps(out), flush=True)
    # a fallback is never silent: the line is out, the status says what it was measured with
    fb = (r_split or {}).get("collective_fallback_reason") if world > 1 else None
    if fb and fb != "asked for" and not args.allow_gather_fallback:
        if rank == 0:
            print("bench.py: the library's collective (gd_gather_export) was NOT used -- %s; the line above was measured with torch.distributed's "
                  "gather.  Exit status 3 (pass --allow-gather-fallback to accept that)." % fb, file=sys.stderr, flush=True)
        sys.exit(3)


if __name__ == "__main__":
    main()
