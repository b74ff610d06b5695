#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_bamdecode.py tests/test_gpu_cli.py -x -q -m gpu > gpurun_out/y_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/y_tests.log

GOLEFT_INGEST_TIMING=1 timeout 300 python tools/scope3.py --length 80000000,63025520,48129895,16571,5000000 > gpurun_out/y_scope3_multi.json 2> gpurun_out/y_scope3_multi.err; grep ingest_list gpurun_out/y_scope3_multi.err | tail -2; python - <<'PY'
import json
d=json.load(open("gpurun_out/y_scope3_multi.json"))
print({k: (round(v["wall_s"],3), v["phases"]) for k, v in d.items() if k.endswith("_decoder")}, d["outputs_identical"])
PY
