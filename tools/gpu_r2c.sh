#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
python -m pytest tests/test_gpu_multidevice.py tests/test_gpu_cli.py tests/test_gpu_ref_fixtures.py tests/test_seqstats.py tests/test_multidepth.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r2c_tests.log
cat gpurun_out/r2c_tests.log
