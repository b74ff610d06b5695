#!/bin/bash
# the device BAM read: CLI / decoder tests, then the reference's own timed invocation on a synthetic 30x chr1 BAM
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
echo "== pytest gpu (CLI, fixtures, BAM decode, multidevice, multidepth)"; timeout 900 python -m pytest tests/test_gpu_cli.py tests/test_gpu_ref_fixtures.py tests/test_gpu_bamdecode.py tests/test_gpu_multidevice.py tests/test_multidepth.py -m gpu -x -q 2>&1 | tail -3
GOLEFT_INGEST_TIMING=1 timeout 900 python tools/scope3.py --paper --name chr1 --length 249250621 2>gpurun_out/scope3_t.err | tail -1 | tee gpurun_out/scope3_chr1.json | cut -c1-700
grep -v amdgpu.ids gpurun_out/scope3_t.err | tail -4
