#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_attn_h2.py tests/test_gpu_comm.py tests/test_gpu_coresidency.py tests/test_gpu_e2e.py tests/test_gpu_gemm_pl.py -x -q 2>&1 | tail -30 | cut -c1-600 | tee $OUT/r06_tail_debug2.txt
