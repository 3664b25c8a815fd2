#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05c24; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_lm.py -q 2>&1 | grep -v amdgpu.ids | tail -2
LM_ARMS="lm_ksliced=1 lm_ksliced=0 lm_ksliced=1 lm_ksliced=0" bash tools/r05/call22.sh
