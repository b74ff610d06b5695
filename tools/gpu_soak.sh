#!/bin/bash
# soak replay: tools/gpu_soak.sh <seed> <jobs> [only]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
GOLEFT_SOAK_VERBOSE=1 GOLEFT_SOAK_JOBS=$2 GOLEFT_SOAK_ONLY=$3 timeout 600 python -m pytest "tests/test_gpu_soak.py::test_soak_one_context_many_jobs[$1]" -q -s 2>&1 | grep "^soak\|fault\|passed\|failed\|Error\|assert" | tail -40
