#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05c16; mkdir -p $O
timeout 600 python tools/r05/plt_stress4.py > $O/plt_stress4.txt 2>&1
tail -60 $O/plt_stress4.txt | cut -c1-900
