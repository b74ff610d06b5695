#!/bin/bash
# One GPU-box session (run through gpurun): stages given as arguments, in order; logs and JSON lines under gpurun_out/.
#   tools/gpu.sh <tag> [tests [pytest args]] [bench:<workload>[:extra bench args]] [prof:<workload>] [cmd:<shell command>] ...
# e.g. tools/gpu.sh r03a tests bench:wgs bench:chr20 "bench:ont:--steps 5 --warmup 2" prof:wgs
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
T=$1; shift
mkdir -p gpurun_out
LOG=gpurun_out/${T}.log
: > $LOG
summ='import sys,json
d=json.loads(sys.stdin.read())
o=d.get("compute_only") or {}
i=d.get("roofline_ingest") or {}
print("  step %.3f ms  value %.3e  kernel %s %.3f ms frac %.3f | compute_only %s ms frac %s (%s) | normalise kernels %s ms wall %s ms frac %s | verified %s" % (
 d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["avg_kernel_ms"], d["roofline"]["frac"],
 o.get("ms_per_step"), (o.get("roofline") or {}).get("frac"), (o.get("roofline") or {}).get("kernel"),
 i.get("kernels_ms"), i.get("wall_ms"), i.get("frac"), d.get("verified_bit_exact")))
print("  kernels_ms", d["kernels_ms"], "slow tiles", d["config"].get("tiles_on_the_generic_kernel"))
for k in ("host_stream_scope", "host_stream_scope_wgs"):
    h = d.get(k)
    if isinstance(h, dict) and "variants" in h:
        print("  %s: %s" % (k, {n: "%.3e ref-b/s %.1f ms (feed %.1f ms) %.1f GB/s" % (v["value"], v["ms"], v["feed_ms"], v["host_to_device_GBps"]) for n, v in h["variants"].items()}))
e = (d.get("emulated_sharding") or {}).get("by_n_gpus") or {}
if e: print("  emulated shards:", {n: "%.2fx (max %.3f ms)" % (v["projected_speedup"], v["max_shard_ms"]) for n, v in e.items()})'
for st in "$@"; do
  case "$st" in
    tests*)
      what=${st#tests}; case "$what" in *tests/*) ;; *) what="tests $what";; esac      # "tests" = the whole directory; "tests tests/x.py -k y" = just that
      echo "== pytest -m gpu $what" >> $LOG
      timeout 1500 python -X faulthandler -m pytest -m gpu --maxfail=8 -q $what > gpurun_out/${T}_pytest.txt 2>&1
      grep -n "Fatal\|fault\|tests/.*line\|passed\|failed\|Error" gpurun_out/${T}_pytest.txt | head -30 >> $LOG
      tail -5 gpurun_out/${T}_pytest.txt >> $LOG ;;
    bench:*)
      w=${st#bench:}; extra=""; case "$w" in *:*) extra=${w#*:}; w=${w%%:*};; esac
      echo "== bench $w $extra" >> $LOG
      timeout 900 python bench.py --workload $w $extra 2>gpurun_out/${T}_$w.err | tail -1 > gpurun_out/${T}_bench_$w.json
      python -c "$summ" < gpurun_out/${T}_bench_$w.json >> $LOG 2>&1 || tail -5 gpurun_out/${T}_$w.err >> $LOG ;;
    prof:*)
      w=${st#prof:}; extra=""; case "$w" in *:*) extra=${w#*:}; w=${w%%:*};; esac
      echo "== prof $w $extra" >> $LOG
      bash tools/prof.sh ${T}_$w --workload $w $extra >> $LOG 2>&1 ;;
    cmd:*)
      echo "== ${st#cmd:}" >> $LOG
      timeout 1500 bash -c "${st#cmd:}" >> $LOG 2>&1 ;;
  esac
done
cat $LOG
