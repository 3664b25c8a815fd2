#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
OMNITOK_TEST_PL_MIN_TOKENS=12288 timeout 1200 python -m pytest tests/test_gpu_temporal_fused.py tests/test_torch_engine.py -m gpu -q --timeout 900 2>&1 | tail -4 | tee -a $OUT/r06_gpu_tests_min_tokens_12288.txt
timeout 600 python -m pytest tests/test_gpu_temporal_fused.py -m gpu -q 2>&1 | tail -2
