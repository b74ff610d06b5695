cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_longread.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "config5" 2>&1 | tail -3
for i in 1 2; do
python bench.py --no-cpu-baseline --no-host-stream --bam-scope off --other-workloads= --emulate-shards= --steps 5 --warmup 2 --workload ont 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('ont: step ms', round(d['ms_per_step'],2), 'frac', round(r['frac'],3), {k: round(v['ms'],2) for k,v in r['kernels'].items()}, 'first', (d.get('first_compute') or {}).get('ms'))"
done
