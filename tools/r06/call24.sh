#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OMNITOK_CORESIDENCY_STEPS=3000 timeout 1500 python -m pytest tests/test_gpu_coresidency.py -q 2>&1 | tail -3 | tee gpurun_out/r06_coresidency_3000.txt
