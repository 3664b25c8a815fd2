#!/bin/bash
# full GPU suite -> gpurun_out/r06_gpu_tests_<tag>.txt
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -q -x --timeout 900 2>&1 | tail -40 > $OUT/r06_gpu_tests_${1:-x}.txt
tail -15 $OUT/r06_gpu_tests_${1:-x}.txt
