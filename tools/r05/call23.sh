#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05c23; mkdir -p $O
timeout 300 python tools/r05/lm_timeline.py > $O/timeline.txt 2>&1
grep -v amdgpu $O/timeline.txt | tail -40
