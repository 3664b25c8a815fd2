#!/bin/bash
# the GPU suite with small calls routed to the fp32-activation flow (the "pl_min_tokens" option's arm): is the option still healthy?
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
OMNITOK_TEST_PL_MIN_TOKENS=12288 timeout 2400 python -m pytest tests -m gpu -q -x --timeout 900 2>&1 | tail -12 > $OUT/r06_gpu_tests_min_tokens_12288.txt
tail -6 $OUT/r06_gpu_tests_min_tokens_12288.txt
