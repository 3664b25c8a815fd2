#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05c24; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_lm.py -q -x 2>&1 | grep -v amdgpu.ids | tail -3
LM_ARMS="lm_attn_short=1 lm_attn_short=0 lm_attn_short=1 lm_attn_short=0" bash tools/r05/call22.sh
